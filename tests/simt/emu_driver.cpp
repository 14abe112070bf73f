// tests/simt/emu_driver.cpp — TEST INFRASTRUCTURE ONLY.
// Runs the product's kernels (zstd_amd/csrc/zhip_kernels.h, unmodified) on the host SIMT emulator so that the
// wave-level logic can be checked against the oracle without a GPU.  Built by tests/_libs.py with g++.
#include <hip/hip_runtime.h>
#include "zhip_kernels.h"

extern "C" {

// parse `nUnits` units with the strategy-fast kernel. seqs: nUnits*ZHIP_SEQ_CAP records, metas: nUnits
void emu_parse_fast(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, ZhipSeq* seqs, uint8_t* lits, ZhipParse* metas,
                    uint32_t smemBytes, int osThreads)
{
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, smemBytes,
                 [=] { zhip::k_parse_fast(src, units, nUnits, seqs, lits, metas); }, osThreads);
}

void emu_parse_dfast(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, uint32_t* tabs, size_t tabStride,
                     ZhipSeq* seqs, uint8_t* lits, ZhipParse* metas, int osThreads)
{
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, zhip::dfast_lds_bytes(),
                 [=] { zhip::k_parse_dfast(src, units, nUnits, tabs, tabStride, seqs, lits, metas); }, osThreads);
}
uint64_t emu_dfast_table_bytes(uint32_t hashLog, uint32_t chainLog) { return zhip::dfast_table_bytes(hashLog, chainLog); }

// hash-chain strategies (greedy / lazy / lazy2): the three launches of zhip_parse_lazy.h
void emu_parse_lazy(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, uint32_t* tabs, size_t tabStride, uint64_t* best,
                    ZhipSeq* seqs, uint8_t* lits, ZhipParse* metas, int osThreads)
{
    uint32_t maxLen = 1, maxHlog = 6;
    for (uint32_t i = 0; i < nUnits; i++) { if (units[i].srcLen > maxLen) maxLen = units[i].srcLen; if (units[i].hashLog > maxHlog) maxHlog = units[i].hashLog; }
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, zhip::hc_chain_lds_bytes(maxHlog),
                 [=] { zhip::k_hc_chain(src, units, nUnits, tabs, tabStride); }, osThreads);
    uint32_t const bpu = (maxLen + ZHIP_HC_SEARCH_THREADS - 1) / ZHIP_HC_SEARCH_THREADS;
    simt::launch({((nUnits + 7) / 8) * 8 * bpu, 1, 1}, {ZHIP_HC_SEARCH_THREADS, 1, 1}, 0,
                 [=] { zhip::k_hc_search(src, units, nUnits, bpu, tabs, tabStride, best); }, osThreads);
    simt::launch({nUnits, 1, 1}, {64, 1, 1}, 0,
                 [=] { zhip::k_parse_lazy(src, units, nUnits, tabs, tabStride, best, seqs, lits, metas); }, osThreads);
}
uint64_t emu_hc_table_words(uint32_t hashLog) { return zhip::hc_table_words(hashLog); }

// stage 2 for `nUnits` units: out slots of ZHIP_OUT_STRIDE bytes, outSize[nUnits]
void emu_entropy(const uint8_t* src, const ZhipUnit* units, uint32_t nUnits, const ZhipSeq* seqs, const ZhipParse* metas,
                 const uint8_t* lits, uint16_t* stBits, uint8_t* out, uint32_t* outSize, int osThreads)
{
    simt::launch({nUnits, 1, 1}, {ZHIP_ENT_THREADS, 1, 1}, sizeof(zhip::EntShared),
                 [=] { zhip::k_entropy(src, units, nUnits, seqs, metas, lits, stBits, out, outSize); }, osThreads);
}
uint32_t emu_out_stride(void) { return ZHIP_OUT_STRIDE; }
uint32_t emu_lit_stride(void) { return ZHIP_LIT_STRIDE; }
uint32_t emu_ent_shared(void) { return (uint32_t)sizeof(zhip::EntShared); }

uint32_t emu_fast_lds_bytes(uint32_t hlog) { return zhip::fast_lds_bytes(hlog); }
uint32_t emu_seq_cap(void) { return ZHIP_SEQ_CAP; }
uint32_t emu_sizeof_unit(void) { return sizeof(ZhipUnit); }
uint32_t emu_sizeof_parse(void) { return sizeof(ZhipParse); }

}
