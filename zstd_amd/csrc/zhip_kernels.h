// zhip_kernels.h — the __global__ entry points (gfx950).  Launch code lives in zhip_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_parse.h"

namespace zhip {

// Stage 1: one wavefront (= one 64-thread workgroup) per unit.  Dynamic LDS = 4 << hashLog bytes (hash table).
__global__ void __launch_bounds__(64)
k_parse_fast(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, uint32_t nUnits,
             ZhipSeq* __restrict__ seqs, ZhipParse* __restrict__ metas)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    parse_fast_unit(src + u.srcOff, u.srcLen, u, (uint32_t*)smem, seqs + (size_t)ui * ZHIP_SEQ_CAP, metas + ui);
}

}  // namespace zhip
