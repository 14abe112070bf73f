// zhip_kernel_params.h — launch parameters both sides need: the kernels (zhip_kernels_*.h) and the host code that sizes their launches (zhip_lib.hip)
#pragma once
#include <stdint.h>

#ifndef ZHIP_COST_SAMPLE
#define ZHIP_COST_SAMPLE 1024u      /* bytes per sample of k_order_cost (four samples per unit); 4096 until round 6: the estimate orders the queue as well from a quarter of the bytes (profiles/r06_ab_fast_cost_sample.log) */
#endif
#define ZHIP_DICT_TICKET 8u
#define ZHIP_XXH_WAVE_LDS (2u * 512u * 8u)
#define ZHIP_SCAN_TILE 4096u

// dynamic LDS of the stage-test hooks k_test_huf / k_test_fse
#include "zhip_tables.h"
namespace zhip {
struct ZhipTestHufShared { HufWork w; uint32_t count[256]; uint32_t code[256]; uint8_t hdr[136]; };
struct ZhipTestFseShared { FseCTable ct; uint32_t count[64]; int16_t norm[64]; uint32_t words[16]; uint8_t ncount[64]; uint8_t cellSym[4096]; uint16_t first[64]; };
}
