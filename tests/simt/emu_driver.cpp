// tests/simt/emu_driver.cpp — TEST INFRASTRUCTURE ONLY.
// Runs the product's kernels (zstd_amd/csrc/zhip_kernels.h, unmodified) on the host SIMT emulator so that the
// wave-level logic can be checked against the oracle without a GPU.  Built by tests/_libs.py with g++.
#include <hip/hip_runtime.h>
#include "zhip_kernels.h"

extern "C" {

// parse `nUnits` units with the strategy-fast kernel. seqs: nUnits*ZHIP_SEQ_CAP records, metas: nUnits
void emu_parse_fast(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, ZhipSeq* seqs, ZhipParse* metas,
                    uint32_t smemBytes, int osThreads)
{
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, smemBytes,
                 [=] { zhip::k_parse_fast(src, units, nUnits, seqs, metas); }, osThreads);
}

uint32_t emu_seq_cap(void) { return ZHIP_SEQ_CAP; }
uint32_t emu_sizeof_unit(void) { return sizeof(ZhipUnit); }
uint32_t emu_sizeof_parse(void) { return sizeof(ZhipParse); }

}
