// zhip_kernels_lazy.h — __global__ entry points: stage 1 for the strategies greedy / lazy / lazy2 on units (hash chain and row hash).
// Compiled into its own code object by zhip_k_lazy.hip: a change in another kernel family cannot move this one's inlining or register allocation
// (round 3 ended on a decoder whose code the block-parallel decoder's arrival had reshaped).  Declarations for the host side: zhip_kernel_decls.h.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_kernel_params.h"
#include "zhip_parse.h"
#include "zhip_parse_lazy.h"

// register caps for more resident wavefronts (A/B-measured): empty = the compiler's own choice
#ifndef ZHIP_DFAST_OCC
#define ZHIP_DFAST_OCC __attribute__((amdgpu_waves_per_eu(4)))   /* with the window (131 VGPRs as compiled): 4 waves per SIMD, A/B on 2 GiB: 3 / 4 / 5 / 6 -> text 195 / 169 / 188 / 252 ms */
#endif
#ifndef ZHIP_LAZY_OCC
#define ZHIP_LAZY_OCC
#endif
#ifndef ZHIP_ENT_OCC
#define ZHIP_ENT_OCC
#endif

namespace zhip {

// Stage 1 for strategies greedy / lazy / lazy2 (hash chain), three launches — see zhip_parse_lazy.h.
// tabs + ui * tabStride words: prev[ZHIP_UNIT_MAX] (row matcher: + its row lists, hc_table_words); best + ui * ZHIP_UNIT_MAX records.
// k_hc_chain: dynamic LDS = hc_chain_lds_bytes(max hashLog).
__global__ void __launch_bounds__(64)
k_hc_chain(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, uint32_t nUnits,
           uint32_t* __restrict__ tabs, size_t tabStride, uint64_t* __restrict__ best /* used as scratch here */)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    if (u.strategy < ZHIP_STRAT_GREEDY) return;
    uint32_t* const prev = tabs + (size_t)ui * tabStride;
    uint32_t* const queue = (uint32_t*)(best + (size_t)ui * ZHIP_UNIT_MAX);
    const uint8_t* const p = src + u.srcOff;
    if (u.rowLog) {                                     // row-hash matcher: links keyed by the row index, heads in LDS
        switch (u.minMatch) {
        case 5:  rh_chain_unit<5>(p, u.srcLen, u, smem, prev); break;
        case 6: case 7: case 8: rh_chain_unit<6>(p, u.srcLen, u, smem, prev); break;
        default: rh_chain_unit<4>(p, u.srcLen, u, smem, prev); break;
        }
        return;
    }
    switch (u.minMatch) {                               // zstd_lazy.c:1531 mls = BOUNDED(4, minMatch, 6)
    case 5:  hc_chain_unit<5>(p, u.srcLen, u, smem, prev, queue); break;
    case 6: case 7: case 8: hc_chain_unit<6>(p, u.srcLen, u, smem, prev, queue); break;
    default: hc_chain_unit<4>(p, u.srcLen, u, smem, prev, queue); break;
    }
}

// one thread per position; workgroup b works on unit (b / 8 / blocksPerUnit) * 8 + b % 8 — consecutive workgroup ids
// go round-robin over the 8 XCDs, so all workgroups of one unit land on the same XCD and share its L2
__global__ void __launch_bounds__(ZHIP_HC_SEARCH_THREADS)
k_hc_search(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, uint32_t nUnits, uint32_t blocksPerUnit,
            const uint32_t* __restrict__ tabs, size_t tabStride, uint64_t* __restrict__ best)
{
    uint32_t const b = blockIdx.x, xcd = b & 7, slot = b >> 3;
    uint32_t const ui = (slot / blocksPerUnit) * 8 + xcd, chunk = slot % blocksPerUnit;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    uint32_t const n = u.srcLen, p = chunk * ZHIP_HC_SEARCH_THREADS + threadIdx.x;
    if (u.strategy < ZHIP_STRAT_GREEDY || n < 10 || p > n - 8) return;
    const uint32_t* const prev = tabs + (size_t)ui * tabStride;
    best[(size_t)ui * ZHIP_UNIT_MAX + p] = hc_search_pos(src + u.srcOff, n, p, prev, u.searchLog, u.chainLog);
}

// k_hc_search with the unit staged in LDS: one 1024-thread workgroup per unit, dynamic LDS = longest unit + 16.
__global__ void __launch_bounds__(ZHIP_HC_SEARCH_LDS_THREADS)
k_hc_search_lds(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, uint32_t nUnits,
                const uint32_t* __restrict__ tabs, size_t tabStride, uint64_t* __restrict__ best,
                const ZhipParse* __restrict__ metas /* not nullptr: only the units whose TRY parse gave up (ZHIP_PARSE_REDO) */)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const ui = blockIdx.x, t = threadIdx.x;
    if (ui >= nUnits) return;
    if (metas && metas[ui].status != ZHIP_PARSE_REDO) return;
    ZhipUnit const u = units[ui];
    uint32_t const n = u.srcLen;
    if (u.strategy < ZHIP_STRAT_GREEDY || n < 10) return;
    const uint8_t* const p0 = src + u.srcOff;
    lds_u8* const lsrc = (lds_u8*)(uintptr_t)smem;
    uint32_t const full = n & ~15u;
    for (uint32_t i = 16u * t; i < full; i += 16u * ZHIP_HC_SEARCH_LDS_THREADS) {
        uint4 v; __builtin_memcpy(&v, p0 + i, 16);
        lds_u32* const d = (lds_u32*)(lsrc + i);                               // 16-byte aligned: merged into one ds_write_b128
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    if (t < 32) { uint32_t const i = full + t; lsrc[i] = i < n ? p0[i] : 0; }      // ragged tail + 16 zero bytes of padding
    __syncthreads();
    const uint32_t* const prev = tabs + (size_t)ui * tabStride;
    uint64_t* const b = best + (size_t)ui * ZHIP_UNIT_MAX;
    if (u.rowLog) { for (uint32_t p = t; p <= n - 8; p += ZHIP_HC_SEARCH_LDS_THREADS) b[p] = rh_search_pos_lds(lsrc, n, p, prev, u.searchLog, u.rowLog, metas != nullptr); }
    else for (uint32_t p = t; p <= n - 8; p += ZHIP_HC_SEARCH_LDS_THREADS) b[p] = hc_search_pos_lds(lsrc, n, p, prev, u.searchLog, u.chainLog);
}

__global__ void __launch_bounds__(64) ZHIP_LAZY_OCC
k_parse_lazy(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
             uint32_t* __restrict__ tabs, size_t tabStride, const uint64_t* __restrict__ best,
             ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas,
             uint32_t mode /* 0: the parse.  The row matcher's two-pass prediction (zhip_parse_lazy.h: rh_reconcile): 2 = TRY — the parse, given up (status
                              ZHIP_PARSE_REDO) once `budget` searches had to be redone live; then, for those units only, 1 = the predicting parse, and 3 = the parse again */,
             uint32_t budget)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)                // ZHIP_RH_DIRTY_BYTES: the row matcher's dirty-row bits
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    if (u.strategy < ZHIP_STRAT_GREEDY) return;
    if ((mode == 1 || mode == 3) && metas[ui].status != ZHIP_PARSE_REDO) return;
    uint32_t* const prev = tabs + (size_t)ui * tabStride;
    parse_lazy_unit(src + u.srcOff, u.srcLen, u, smem, prev, best + (size_t)ui * ZHIP_UNIT_MAX,
                    seqs + slots[ui].seqOff, lits + slots[ui].litOff, metas + ui, mode == 1, mode == 2 ? budget : 0u, mode == 3);
}

}  // namespace zhip
