// zhip_kernels.h — the __global__ entry points (gfx950).  Launch code lives in zhip_launch.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_parse.h"
#include "zhip_parse_dfast.h"
#include "zhip_parse_lazy.h"
#include "zhip_parse_dict.h"
#include "zhip_parse_ext.h"
#include "zhip_parse_lane.h"
#include "zhip_entropy.h"
#include "zhip_frame.h"
#include "zhip_frame_lazy.h"
#include "zhip_decode.h"
#include "zhip_decode_big.h"

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

// Stage 1: one wavefront (= one 64-thread workgroup) per unit.  Dynamic LDS = fast_lds_bytes(hashLog).
#ifndef ZHIP_FAST_OCC
#define ZHIP_FAST_OCC __attribute__((amdgpu_waves_per_eu(3)))      /* <= 170 VGPRs: the LDS table admits nine units per CU = three on one of the four SIMDs (left alone the compiler has chosen anything from 141 to 248) */
#endif
__global__ void __launch_bounds__(64) ZHIP_FAST_OCC
k_parse_fast(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
             ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    if (u.strategy != ZHIP_STRAT_FAST || u.pad1 == ZHIP_UNIT_LANE) return;          // another family's kernel handles it
    const uint8_t* const p = src + u.srcOff;
    ZhipSlot const sl = slots[ui];
    ZhipSeq* const sq = seqs + sl.seqOff;
    uint8_t* const lt = lits + sl.litOff;
    switch (u.minMatch) {               // wave-uniform: the hash width is a compile-time constant inside the parser
    case 5:  parse_fast_unit<5>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
    case 6:  parse_fast_unit<6>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
    case 7:  parse_fast_unit<7>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
    case 8:  parse_fast_unit<8>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
    default: parse_fast_unit<4>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
    }
}

// Stage 1, queue form: persistent wavefronts take units from a ticket counter, in the order `order[]` gives (heaviest first, k_order_*;
// nullptr = as they come).  Two kernels share ONE queue: k_parse_fast_q keeps its table in LDS (nine wavefronts fill a CU's LDS),
// k_parse_fast_g keeps it in global memory and needs no LDS at all, so its wavefronts run BESIDE the nine on the same CU and hide the
// latency those cannot.  Whoever is free takes the next unit: the split between the two adjusts itself.
__device__ __forceinline__ uint32_t queue_take(uint32_t* queue)
{
    uint32_t t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(queue, 1u);
    return __builtin_amdgcn_readfirstlane(t);
}
#ifndef ZHIP_FASTG_OCC
#define ZHIP_FASTG_OCC
#endif
__global__ void __launch_bounds__(64) ZHIP_FAST_OCC
k_parse_fast_q(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
               ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas,
               const uint32_t* __restrict__ order, uint32_t* __restrict__ queue)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    for (;;) {
        uint32_t const t = queue_take(queue);
        if (t >= nUnits) return;
        uint32_t const ui = order ? order[t] : t;
        ZhipUnit const u = units[ui];
        if (u.strategy != ZHIP_STRAT_FAST || u.pad1 == ZHIP_UNIT_LANE) continue;
        const uint8_t* const p = src + u.srcOff;
        ZhipSlot const sl = slots[ui];
        ZhipSeq* const sq = seqs + sl.seqOff;
        uint8_t* const lt = lits + sl.litOff;
        switch (u.minMatch) {
        case 5:  parse_fast_unit<5>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
        case 6:  parse_fast_unit<6>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
        case 7:  parse_fast_unit<7>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
        case 8:  parse_fast_unit<8>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
        default: parse_fast_unit<4>(p, u.srcLen, u, smem, sq, lt, metas + ui); break;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ void __launch_bounds__(64) ZHIP_FASTG_OCC
k_parse_fast_g(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
               ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas,
               const uint32_t* __restrict__ order, uint32_t* __restrict__ queue, uint32_t* __restrict__ gtabs, uint32_t gtabWords)
{
    uint32_t* const gtab = gtabs + (size_t)blockIdx.x * gtabWords;
    for (;;) {
        uint32_t const t = queue_take(queue);
        if (t >= nUnits) return;
        uint32_t const ui = order ? order[t] : t;
        ZhipUnit const u = units[ui];
        if (u.strategy != ZHIP_STRAT_FAST || u.pad1 == ZHIP_UNIT_LANE) continue;
        const uint8_t* const p = src + u.srcOff;
        ZhipSlot const sl = slots[ui];
        ZhipSeq* const sq = seqs + sl.seqOff;
        uint8_t* const lt = lits + sl.litOff;
        switch (u.minMatch) {
        case 5:  parse_fast_unit_g<5>(p, u.srcLen, u, gtab, sq, lt, metas + ui); break;
        case 6:  parse_fast_unit_g<6>(p, u.srcLen, u, gtab, sq, lt, metas + ui); break;
        case 7:  parse_fast_unit_g<7>(p, u.srcLen, u, gtab, sq, lt, metas + ui); break;
        case 8:  parse_fast_unit_g<8>(p, u.srcLen, u, gtab, sq, lt, metas + ui); break;
        default: parse_fast_unit_g<4>(p, u.srcLen, u, gtab, sq, lt, metas + ui); break;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Dispatch order for the queue kernels: units sorted by descending cost (a counting sort over 2 048 cost classes, one workgroup).
// cost: mode 2 = the sequence count the previous call left in metas[] (measurement only: the upper bound an estimator can reach),
// mode 1 = k_order_cost's estimate.
__global__ void __launch_bounds__(1024)
k_order_sort(const uint32_t* __restrict__ cost, uint32_t nUnits, uint32_t* __restrict__ order)
{
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t part[1024];
    uint32_t const tid = threadIdx.x;
    hist[tid] = 0; hist[tid + 1024] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < nUnits; i += 1024) { uint32_t const b = cost[i] >> 4; atomicAdd(&hist[2047u - (b < 2047u ? b : 2047u)], 1u); }
    __syncthreads();
    // exclusive prefix over the classes (class 0 = the most expensive): two classes per thread, then a scan of the pair sums
    uint32_t const a0 = hist[2 * tid], a1 = hist[2 * tid + 1];
    part[tid] = a0 + a1;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t const v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t const base = part[tid] - (a0 + a1);
    hist[2 * tid] = base; hist[2 * tid + 1] = base + a0;
    __syncthreads();
    for (uint32_t i = tid; i < nUnits; i += 1024) { uint32_t const b = cost[i] >> 4; order[atomicAdd(&hist[2047u - (b < 2047u ? b : 2047u)], 1u)] = i; }
}
// cost estimate of a ZSTD_fast unit = its expected number of sequences: four 4 KB samples are scanned densely against a table of 16-bit
// TAGS (a second hash of the 4 bytes the parser compares) — a lane "hits" when the slot of its hash holds its own tag, and a run of
// hitting lanes is one match.  Coalesced source reads only, no candidate fetch: ~1 % of the parse it schedules.
#define ZHIP_COST_SAMPLE 4096u
__global__ void __launch_bounds__(64)
k_order_cost(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, uint32_t nUnits, uint32_t* __restrict__ cost)
{
    __shared__ uint16_t tags[8192];
    uint32_t const ui = blockIdx.x, lane = threadIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    uint32_t const n = u.srcLen;
    if (n < 4u * ZHIP_COST_SAMPLE + 16u) { if (lane == 0) cost[ui] = n >> 5; return; }
    for (uint32_t i = lane; i < 4096; i += 64) ((uint32_t*)tags)[i] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint8_t* const p = src + u.srcOff;
    uint32_t const stride = (n - ZHIP_COST_SAMPLE - 16u) / 3u;
    uint32_t const mls = u.minMatch;
    uint32_t runs = 0;
    for (uint32_t r = 0; r < 4; r++) {
        uint32_t const base = r * stride;
        uint32_t carry = 1;                                                   // a sample's first lane does not open a run
        for (uint32_t w = 0; w < ZHIP_COST_SAMPLE; w += 256) {
            uint64_t c[4];
            #pragma unroll
            for (int k = 0; k < 4; k++) c[k] = zhip::ld64(p + base + w + 64u * k + lane);
            #pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t const h = (mls <= 4 ? zhip::hash_pos<4>(c[k], 19) : mls == 5 ? zhip::hash_pos<5>(c[k], 19) : zhip::hash_pos<6>(c[k], 19));
                uint16_t const tag = (uint16_t)((((uint32_t)c[k] * 2246822519u) >> 16) | 1u);
                uint16_t const oldTag = tags[h];
                __builtin_amdgcn_wave_barrier();
                tags[h] = tag;
                __builtin_amdgcn_wave_barrier();
                unsigned long long const H = __ballot(oldTag == tag);
                runs += (uint32_t)__builtin_popcountll(H & ~((H << 1) | carry));
                carry = (uint32_t)(H >> 63);
            }
        }
    }
    if (lane == 0) cost[ui] = runs * (n / (4u * ZHIP_COST_SAMPLE)) + (n >> 7);
}
__global__ void k_order_cost_stale(const ZhipParse* __restrict__ metas, uint32_t nUnits, uint32_t* __restrict__ cost)
{
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nUnits) cost[i] = metas[i].nbSeq;
}

// Stage 1 for strategy dfast: one wavefront per unit, the unit's two hash tables live in HBM/L2 (tabs + ui * tabStride
// words: long table, then short table).  Dynamic LDS = dfast_lds_bytes().
__global__ void __launch_bounds__(64) ZHIP_DFAST_OCC
k_parse_dfast(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
              uint32_t* __restrict__ tabs, size_t tabStride,
              ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    // persistent workgroups: workgroup w takes the units w, w + gridDim.x, ... and reuses ONE table pair (tabs + w * tabStride) for
    // all of them, so the table memory in use is gridDim.x pairs, not nUnits pairs
    for (uint32_t ui = blockIdx.x; ui < nUnits; ui += gridDim.x) {
        ZhipUnit const u = units[ui];
        if (u.strategy != ZHIP_STRAT_DFAST || u.pad1 == ZHIP_UNIT_LANE) continue;
        const uint8_t* const p = src + u.srcOff;
        ZhipSlot const sl = slots[ui];
        ZhipSeq* const sq = seqs + sl.seqOff;
        uint8_t* const lt = lits + sl.litOff;
        uint32_t* const tL = tabs + (size_t)blockIdx.x * tabStride;
        uint32_t* const tS = tL + ((size_t)1 << u.hashLog);
        switch (u.minMatch) {
        case 5:  parse_dfast_unit<5>(p, u.srcLen, u, smem, tL, tS, sq, lt, metas + ui); break;
        case 6:  parse_dfast_unit<6>(p, u.srcLen, u, smem, tL, tS, sq, lt, metas + ui); break;
        case 7:  parse_dfast_unit<7>(p, u.srcLen, u, smem, tL, tS, sq, lt, metas + ui); break;
        case 8:  parse_dfast_unit<8>(p, u.srcLen, u, smem, tL, tS, sq, lt, metas + ui); break;
        default: parse_dfast_unit<4>(p, u.srcLen, u, smem, tL, tS, sq, lt, metas + ui); break;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// Stage 1 for LARGE batches (zhip_parse_lane.h): one LANE per unit, tables (zeroed by the host's memset) at tabs + ui * tabStride words
__global__ void __launch_bounds__(64)
k_parse_lane(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
             uint32_t* __restrict__ tabs, size_t tabStride, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas)
{
    uint32_t const ui = blockIdx.x * 64u + threadIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    if (u.pad1 != ZHIP_UNIT_LANE) return;
    ZhipSlot const sl = slots[ui];
    parse_lane_unit(src + u.srcOff, u, tabs + (size_t)ui * tabStride, seqs + sl.seqOff, sl.seqCap, lits + sl.litOff, metas + ui);
}

// Stage 1 for records compressed with an attached dictionary (strategies fast and dfast), one wavefront per record.
// Dynamic LDS = max(dict_lds_bytes(hashLog, chainLog), dict_fast_lds_bytes(hashLog)) over the records.
template <bool GLOB>
__device__ __forceinline__ void parse_dict_record(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t ui,
                                                  const ZhipCDictDev& cd, unsigned char* tabmem, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas)
{
    ZhipUnit const u = units[ui];
    if (u.pad0 == ZHIP_UNIT_COPYMODE) return;             // above the attach cut-off: k_parse_ext's
    const uint8_t* const p = src + u.srcOff;
    ZhipSlot const sl = slots[ui];
    if (u.strategy == ZHIP_STRAT_FAST) {
        switch (u.minMatch) {
        case 5:  parse_fast_dms_unit<5, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
        case 6:  parse_fast_dms_unit<6, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
        case 7: case 8: parse_fast_dms_unit<7, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
        default: parse_fast_dms_unit<4, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
        }
        return;
    }
    if (u.strategy != ZHIP_STRAT_DFAST) return;
    switch (u.minMatch) {
    case 5:  parse_dfast_dms_unit<5, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
    case 6:  parse_dfast_dms_unit<6, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
    case 7:  parse_dfast_dms_unit<7, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
    case 8:  parse_dfast_dms_unit<8, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
    default: parse_dfast_dms_unit<4, GLOB>(p, u.srcLen, u, cd, tabmem, seqs + sl.seqOff, lits + sl.litOff, metas + ui); break;
    }
}
__global__ void __launch_bounds__(64)
k_parse_dict(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
             ZhipCDictDev cd, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    parse_dict_record<false>(src, units, slots, ui, cd, smem, seqs, lits, metas);
}
// The same stage as a ticket queue (ZHIP_DICT_TICKET records per ticket: ten million records on one counter): persistent wavefronts
// with the record's tables in LDS (k_parse_dict_q, as many as the LDS admits) and, beside them on the same CUs, persistent wavefronts with
// the tables in a per-wavefront region of global memory (k_parse_dict_g: `gtabs + blockIdx.x * gtabBytes`) — the stage is a chain of
// dependent round trips per match, so what it lacks is wavefronts in flight, and 62 registers admit three times what the LDS does.
#define ZHIP_DICT_TICKET 8u
__global__ void __launch_bounds__(64)
k_parse_dict_q(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
               ZhipCDictDev cd, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas, uint32_t* __restrict__ queue)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    for (;;) {
        uint32_t t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(queue, ZHIP_DICT_TICKET);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= nUnits) return;
        uint32_t const tEnd = t + ZHIP_DICT_TICKET < nUnits ? t + ZHIP_DICT_TICKET : nUnits;
        for (uint32_t ui = t; ui < tEnd; ui++) {
            parse_dict_record<false>(src, units, slots, ui, cd, smem, seqs, lits, metas);
            __builtin_amdgcn_wave_barrier();
        }
    }
}
__global__ void __launch_bounds__(64)
k_parse_dict_g(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
               ZhipCDictDev cd, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas, uint32_t* __restrict__ queue,
               unsigned char* __restrict__ gtabs, uint32_t gtabBytes)
{
    unsigned char* const gtab = gtabs + (size_t)blockIdx.x * gtabBytes;
    for (;;) {
        uint32_t t = 0;
        if ((threadIdx.x & 63) == 0) t = atomicAdd(queue, ZHIP_DICT_TICKET);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= nUnits) return;
        uint32_t const tEnd = t + ZHIP_DICT_TICKET < nUnits ? t + ZHIP_DICT_TICKET : nUnits;
        for (uint32_t ui = t; ui < tEnd; ui++) {
            parse_dict_record<true>(src, units, slots, ui, cd, gtab, seqs, lits, metas);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Stage 1 for strategies greedy / lazy / lazy2 (hash chain), three launches — see zhip_parse_lazy.h.
// tabs + ui * tabStride words: prev[ZHIP_UNIT_MAX]; best + ui * ZHIP_UNIT_MAX records.
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
    if (u.rowLog) { for (uint32_t p = t; p <= n - 8; p += ZHIP_HC_SEARCH_LDS_THREADS) b[p] = rh_search_pos_lds(lsrc, n, p, prev, u.searchLog, u.rowLog); }
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

// Copy mode of a dictionary (sources above the attach cut-off): k_ext_init gives every such source a private copy of the
// CDict's tables with the tags stripped (zstd_compress.c:2379-2393), k_parse_ext runs one source per LANE (zhip_parse_ext.h).
__global__ void __launch_bounds__(256)
k_ext_init(const uint32_t* __restrict__ cdTabL, const uint32_t* __restrict__ cdTabS, uint32_t wordsL, uint32_t wordsS,
           uint32_t* __restrict__ tabs, size_t tabStride)
{
    uint32_t* const t = tabs + (size_t)blockIdx.y * tabStride;
    uint32_t const total = wordsL + wordsS;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) t[i] = (i < wordsL ? cdTabL[i] : cdTabS[i - wordsL]) >> 8;
}
__global__ void __launch_bounds__(64)
k_parse_ext(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, const uint32_t* __restrict__ extIdx,
            uint32_t nExt, ZhipCDictDev cd, uint32_t* __restrict__ tabs, size_t tabStride,
            ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas)
{
    uint32_t const i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nExt) return;
    uint32_t const ui = extIdx[i];
    ZhipUnit const u = units[ui];
    ZhipSlot const sl = slots[ui];
    parse_ext_source(src + u.srcOff, u, cd, tabs + (size_t)i * tabStride, seqs + sl.seqOff, sl.seqCap, lits + sl.litOff, metas + ui);
}

// Stage 2: literals + sequences entropy coding and frame assembly into the unit's output slot.  Two shapes of the same code
// (zhip_entropy.h): one 256-thread workgroup per unit (dynamic LDS = sizeof(EntShared)) and, for units of at most
// ZHIP_ENT_SMALL_MAX bytes, one wavefront per unit (k_entropy_small, sizeof(EntSharedSmall)).  sizeClass: 0 = every unit,
// 1 = only the units above ZHIP_ENT_SMALL_MAX (the small ones belong to the other launch).
__global__ void __launch_bounds__(ZHIP_ENT_THREADS) ZHIP_ENT_OCC
k_entropy(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
          const ZhipSeq* __restrict__ seqs, const ZhipParse* __restrict__ metas,
          const uint8_t* __restrict__ lits, uint16_t* __restrict__ stBits, uint8_t* __restrict__ out, uint32_t* __restrict__ outSize,
          const ZhipDictEntropy* __restrict__ dictEntropy, uint32_t dictID, const uint32_t* __restrict__ checks /* frame checksums or nullptr */,
          uint32_t sizeClass)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    if (sizeClass == 1 && u.srcLen <= ZHIP_ENT_SMALL_MAX) return;
    ZhipParse const pm = metas[ui];
    ZhipSlot const sl = slots[ui];
    entropy_unit<ZHIP_ENT_THREADS, EntShared>(src + u.srcOff, u, seqs + sl.seqOff, pm, lits + sl.litOff,
                 stBits + 3 * sl.seqOff, sl.seqCap, out + sl.outOff, outSize + ui, (EntShared*)smem, dictEntropy, dictID, checks != nullptr, checks ? checks[ui] : 0u);
}
__global__ void __launch_bounds__(64)
k_entropy_small(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
                const ZhipSeq* __restrict__ seqs, const ZhipParse* __restrict__ metas,
                const uint8_t* __restrict__ lits, uint16_t* __restrict__ stBits, uint8_t* __restrict__ out, uint32_t* __restrict__ outSize,
                const ZhipDictEntropy* __restrict__ dictEntropy, uint32_t dictID, const uint32_t* __restrict__ checks)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    if (u.srcLen > ZHIP_ENT_SMALL_MAX) return;
    ZhipParse const pm = metas[ui];
    ZhipSlot const sl = slots[ui];
    entropy_unit<64, EntSharedSmall>(src + u.srcOff, u, seqs + sl.seqOff, pm, lits + sl.litOff,
                 stBits + 3 * sl.seqOff, sl.seqCap, out + sl.outOff, outSize + ui, (EntSharedSmall*)smem, dictEntropy, dictID, checks != nullptr, checks ? checks[ui] : 0u);
}

// One workgroup per multi-block frame (zhip_frame.h).  frames[i].srcLen is the whole input of frame i (< 2^31); its slot gives
// one block's worth of sequence / literal room (reused block after block) and the frame's output room.  Dynamic LDS =
// Dynamic LDS = frame_lds_bytes(largest frame_table_lds_bytes); frames whose table does not fit LDS use tabs + i * tabStride words.
template <int OCC>
__device__ __forceinline__ void frame_kernel_body(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ frames, const ZhipSlot* __restrict__ slots, uint32_t nFrames,
             uint32_t* __restrict__ tabs, size_t tabStride, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits,
             uint16_t* __restrict__ stBits, uint8_t* __restrict__ out, uint32_t* __restrict__ outSize, ZhipFrameState* __restrict__ states,
             const uint32_t* __restrict__ checks, const ZhipJob* __restrict__ jobs)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const fi = blockIdx.x;
    if (fi >= nFrames) return;
    ZhipUnit const u = frames[fi];
    if (u.strategy >= ZHIP_STRAT_GREEDY) return;                     // a frame of the lazy strategies: k_frame_lazy's
    ZhipSlot const sl = slots[fi];
    EntShared* const sh = (EntShared*)smem;
    size_t const shBytes = (sizeof(EntShared) + 15) & ~(size_t)15;
    FrameShared* const fs = (FrameShared*)(smem + shBytes);
    // a job's positions count from the start of its window (the prefix in front of its section), a frame's from the frame start
    const ZhipJob* const job = jobs ? jobs + fi : (const ZhipJob*)nullptr;
    uint32_t const mode = frame_table_mode(u.strategy, u.hashLog, (uint64_t)u.srcLen + (job ? job->prefixLen + 1u : 0u));
    unsigned char* const ltab = smem + shBytes + sizeof(FrameShared);
    WideTab T; Lds24Tab T24;
    T.w = mode == ZHIP_FT_HBM ? tabs + (size_t)fi * tabStride : (uint32_t*)ltab;
    T24.lo = (lds_u16*)(uintptr_t)ltab; T24.hi = (lds_u8*)(uintptr_t)(ltab + (2u << u.hashLog));
    uint32_t const shift = frame_job_shift(job, mode);
    const uint8_t* const p = src + u.srcOff + (job ? (size_t)(job->start - job->prefixLen) : 0u) - shift;
    ZhipSeq* const sq = seqs + sl.seqOff;
    uint8_t* const lt = lits + sl.litOff;
    uint16_t* const sb = stBits + 3 * sl.seqOff;
    uint8_t* const o = out + sl.outOff;
    bool const ck = checks != nullptr; uint32_t const cv = ck ? checks[jobs ? jobs[fi].frameIdx : fi] : 0u;      // jobs: the checksum of the whole frame
    frame_fast<OCC>(p, u, T, T24, mode == ZHIP_FT_LDS24, sq, lt, sb, sl.seqCap, o, outSize + fi, sh, fs, states + fi, ck, cv, job, shift);
}
// launches with a table in LDS (ZSTD_fast, hashLog <= 14): two workgroups per CU (2 x 75 KB of LDS) -> 256 registers per lane
__global__ void __launch_bounds__(ZHIP_ENT_THREADS, 2)
k_frame_fast(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ frames, const ZhipSlot* __restrict__ slots, uint32_t nFrames,
             uint32_t* __restrict__ tabs, size_t tabStride, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits,
             uint16_t* __restrict__ stBits, uint8_t* __restrict__ out, uint32_t* __restrict__ outSize, ZhipFrameState* __restrict__ states,
             const uint32_t* __restrict__ checks, const ZhipJob* __restrict__ jobs /* nullptr: every unit is a whole frame; else unit i is one job of frame jobs[i].frameIdx */)
{
    frame_kernel_body<2>(src, frames, slots, nFrames, tabs, tabStride, seqs, lits, stBits, out, outSize, states, checks, jobs);
}
// launches whose tables all live in HBM (ZSTD_dfast, larger ZSTD_fast tables): LDS is 27 KB per workgroup, so the register file
// decides — four workgroups per CU at 128 registers per lane
__global__ void __launch_bounds__(ZHIP_ENT_THREADS, 4)
k_frame_hbm(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ frames, const ZhipSlot* __restrict__ slots, uint32_t nFrames,
            uint32_t* __restrict__ tabs, size_t tabStride, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits,
            uint16_t* __restrict__ stBits, uint8_t* __restrict__ out, uint32_t* __restrict__ outSize, ZhipFrameState* __restrict__ states,
            const uint32_t* __restrict__ checks, const ZhipJob* __restrict__ jobs)
{
    frame_kernel_body<4>(src, frames, slots, nFrames, tabs, tabStride, seqs, lits, stBits, out, outSize, states, checks, jobs);
}

// jobs -> frames: frameSizes[f] = sum of the compressed sizes of frame f's jobs (frameSizes zeroed by the caller)
__global__ void k_frame_sizes(const uint32_t* __restrict__ outSize, const ZhipJob* __restrict__ jobs, uint32_t nJobs, uint32_t* __restrict__ frameSizes)
{
    uint32_t const i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nJobs) atomicAdd(frameSizes + jobs[i].frameIdx, outSize[i]);
}

// Multi-block frames / jobs of the strategies greedy, lazy, lazy2 (zhip_frame_lazy.h), three launches over the same workgroup-units:
// lz[i] says where unit i's links / tags / records / head table live; jobs as in k_frame_fast (nullptr: whole frames).
// the start of unit i's window in the source
__device__ __forceinline__ const uint8_t* lz_window(const uint8_t* __restrict__ src, const ZhipUnit& u, const ZhipJob* __restrict__ jobs, uint32_t i)
{
    return src + u.srcOff + (jobs ? (size_t)(jobs[i].start - jobs[i].prefixLen) : (size_t)0);
}
// k_lz_links: dynamic LDS = sizeof(LzLinkShared)
__global__ void __launch_bounds__(ZHIP_LZ_LINK_THREADS)
k_lz_links(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipJob* __restrict__ jobs, const ZhipLzSlot* __restrict__ lz, uint32_t nW,
           uint32_t* __restrict__ prev, uint8_t* __restrict__ tags, uint32_t* __restrict__ heads)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const wi = blockIdx.x;
    if (wi >= nW) return;
    ZhipUnit const u = units[wi];
    if (u.strategy < ZHIP_STRAT_GREEDY) return;                      // a ZSTD_fast / ZSTD_dfast frame of a mixed batch: k_frame_fast's
    ZhipLzSlot const L = lz[wi];
    const uint8_t* const p = lz_window(src, u, jobs, wi);
    LzLinkShared* const sh = (LzLinkShared*)smem;
    switch (lz_mls(u)) {
    case 5:  lz_links_t<5>(p, u, L, sh, prev + L.posOff, tags + L.posOff, heads + L.headOff); break;
    case 6:  lz_links_t<6>(p, u, L, sh, prev + L.posOff, tags + L.posOff, heads + L.headOff); break;
    default: lz_links_t<4>(p, u, L, sh, prev + L.posOff, tags + L.posOff, heads + L.headOff); break;
    }
}
// k_lz_search: grid (ceil(longest section / 256), nW); one thread per position of the unit's section
__global__ void __launch_bounds__(256)
k_lz_search(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipJob* __restrict__ jobs, const ZhipLzSlot* __restrict__ lz, uint32_t wBase, uint32_t nW,
            const uint32_t* __restrict__ prev, const uint8_t* __restrict__ tags, LzRec* __restrict__ best)
{
    uint32_t const wi = wBase + blockIdx.y;
    if (wi >= nW) return;
    ZhipUnit const u = units[wi];
    if (u.strategy < ZHIP_STRAT_GREEDY) return;
    ZhipLzSlot const L = lz[wi];
    uint32_t const j0 = jobs ? jobs[wi].prefixLen : 0u;
    uint32_t const p = j0 + blockIdx.x * 256u + threadIdx.x;
    if (L.span < 9 || p > L.span - 8) return;
    const uint8_t* const w = lz_window(src, u, jobs, wi);
    uint32_t const maxDist = 1u << u.windowLog, lowLimit = p > maxDist ? p - maxDist : 0u;
    best[L.posOff + p] = u.rowLog ? lz_search_rh(w, L.span, p, prev + L.posOff, tags + L.posOff, u.searchLog, u.rowLog, lowLimit)
                                  : lz_search_hc(w, L.span, p, prev + L.posOff, u.searchLog, u.chainLog, lowLimit);
}
// k_lz_predict: one wavefront per unit, dynamic LDS = sizeof(ZhipParse): the predicting parse (frame_lazy_predict); k_lz_search runs again after it
__global__ void __launch_bounds__(64)
k_lz_predict(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipJob* __restrict__ jobs, const ZhipLzSlot* __restrict__ lz, uint32_t nW,
             uint32_t* __restrict__ prev, const uint8_t* __restrict__ tags, const LzRec* __restrict__ best)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const wi = blockIdx.x;
    if (wi >= nW) return;
    ZhipUnit const u = units[wi];
    if (u.strategy < ZHIP_STRAT_GREEDY || u.srcLen == 0) return;
    ZhipLzSlot const L = lz[wi];
    frame_lazy_predict(lz_window(src, u, jobs, wi), u, prev + L.posOff, tags + L.posOff, best + L.posOff, (ZhipParse*)smem, jobs ? jobs + wi : (const ZhipJob*)nullptr);
}
// k_frame_lazy: dynamic LDS = frame_lazy_lds_bytes()
__global__ void __launch_bounds__(ZHIP_ENT_THREADS, 2)
k_frame_lazy(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, const ZhipJob* __restrict__ jobs,
             const ZhipLzSlot* __restrict__ lz, uint32_t nW, uint32_t* __restrict__ prev, const uint8_t* __restrict__ tags, const LzRec* __restrict__ best,
             uint32_t* __restrict__ heads, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, uint16_t* __restrict__ stBits,
             uint8_t* __restrict__ out, uint32_t* __restrict__ outSize, ZhipFrameState* __restrict__ states, const uint32_t* __restrict__ checks,
             uint32_t havePred /* k_lz_predict ran before: compare what the parse decides with what it marked */)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const wi = blockIdx.x;
    if (wi >= nW) return;
    ZhipUnit const u = units[wi];
    if (u.strategy < ZHIP_STRAT_GREEDY) return;
    ZhipSlot const sl = slots[wi];
    ZhipLzSlot const L = lz[wi];
    EntShared* const sh = (EntShared*)smem;
    LzFrameShared* const fs = (LzFrameShared*)(smem + ((sizeof(EntShared) + 15) & ~(size_t)15));
    const ZhipJob* const job = jobs ? jobs + wi : (const ZhipJob*)nullptr;
    bool const ck = checks != nullptr; uint32_t const cv = ck ? checks[jobs ? jobs[wi].frameIdx : wi] : 0u;
    frame_lazy(lz_window(src, u, jobs, wi), u, L, prev + L.posOff, tags + L.posOff, best + L.posOff, heads + L.headOff,
               seqs + sl.seqOff, lits + sl.litOff, stBits + 3 * sl.seqOff, sl.seqCap, out + sl.outOff, outSize + wi, sh, fs, states + wi, ck, cv, job, havePred != 0);
}

// Frame checksum (ZSTD_c_checksumFlag): XXH64 of each unit's content, low 32 bits (zstd_compress.c:5297-5303).  XXH64 has four
// independent 64-bit lanes over 32-byte stripes and a strictly sequential round per lane (rotate-multiply, not
// associative), so a unit gets 4 GPU lanes — one per accumulator — and a wavefront hashes 16 units at once; the finish
// (merge, tail bytes, avalanche) runs on the group's first lane.
__device__ __forceinline__ uint64_t xxh_rotl(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t xxh_round(uint64_t acc, uint64_t in) { acc += in * 0xC2B2AE3D27D4EB4FULL; return xxh_rotl(acc, 31) * 0x9E3779B185EBCA87ULL; }
__global__ void __launch_bounds__(64)
k_xxh64(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, uint32_t nUnits, uint32_t* __restrict__ checks)
{
    uint64_t const P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    uint32_t const lane = threadIdx.x & 63, j = lane & 3;
    uint32_t const ui = blockIdx.x * 16 + (lane >> 2);
    bool const on = ui < nUnits;
    ZhipUnit const u = units[on ? ui : 0];
    const uint8_t* const p = src + u.srcOff;
    uint32_t const n = on ? u.srcLen : 0, stripes = n >> 5;
    uint64_t v = j == 0 ? P1 + P2 : (j == 1 ? P2 : (j == 2 ? 0 : 0 - P1));
    uint32_t s = 0;
    for (; s + 4 <= stripes; s += 4) {                      // four loads in flight per lane
        uint64_t a[4];
        for (int q = 0; q < 4; q++) __builtin_memcpy(&a[q], p + 32u * (s + (uint32_t)q) + 8u * j, 8);
        for (int q = 0; q < 4; q++) v = xxh_round(v, a[q]);
    }
    for (; s < stripes; s++) { uint64_t a; __builtin_memcpy(&a, p + 32u * s + 8u * j, 8); v = xxh_round(v, a); }
    // gather the four accumulators on the group's first lane
    uint32_t const g0 = lane & ~3u;
    uint64_t vv[4];
    for (int q = 0; q < 4; q++) {
        uint32_t const lo = __shfl((uint32_t)v, (int)(g0 + (uint32_t)q)), hi = __shfl((uint32_t)(v >> 32), (int)(g0 + (uint32_t)q));
        vv[q] = ((uint64_t)hi << 32) | lo;
    }
    if (j == 0 && on) {
        uint64_t h;
        if (n >= 32) {
            h = xxh_rotl(vv[0], 1) + xxh_rotl(vv[1], 7) + xxh_rotl(vv[2], 12) + xxh_rotl(vv[3], 18);
            for (int q = 0; q < 4; q++) h = (h ^ xxh_round(0, vv[q])) * P1 + P4;
        } else h = P5;
        h += (uint64_t)n;
        uint32_t pos = stripes << 5;
        while (pos + 8 <= n) { uint64_t a; __builtin_memcpy(&a, p + pos, 8); h ^= xxh_round(0, a); h = xxh_rotl(h, 27) * P1 + P4; pos += 8; }
        if (pos + 4 <= n) { uint32_t a; __builtin_memcpy(&a, p + pos, 4); h ^= (uint64_t)a * P1; h = xxh_rotl(h, 23) * P2 + P3; pos += 4; }
        while (pos < n) { h ^= (uint64_t)p[pos++] * P5; h = xxh_rotl(h, 11) * P1; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        checks[ui] = (uint32_t)h;
    }
}

// The same for LARGE units (whole frames of many blocks): one wavefront per unit.  The accumulator round
// v = rotl(v + in * P2, 31) * P1 is serial in v, but in * P2 is not: all 64 lanes fetch 4 KB (coalesced) and pre-multiply it into
// LDS while lanes 0..3 — one per accumulator — run the rotate-multiply chains over the block staged before.  The chain (about three
// quarter-rate 32-bit multiplies per 32 input bytes) is what bounds one frame's checksum; frames of a batch hash side by side.
#define ZHIP_XXH_WAVE_LDS (2u * 512u * 8u)
__global__ void __launch_bounds__(64)
k_xxh64_wave(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, uint32_t nUnits, uint32_t* __restrict__ checks)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint64_t (*prod)[512] = (uint64_t (*)[512])smem;                    // two blocks of 128 stripes x 4 accumulators, pre-multiplied by P2
    uint64_t const P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
    uint32_t const lane = threadIdx.x & 63, j = lane & 3, ui = blockIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    const uint8_t* const p = src + u.srcOff;
    uint32_t const n = u.srcLen, stripes = n >> 5, blocks = stripes >> 7;
    uint64_t v = j == 0 ? P1 + P2 : (j == 1 ? P2 : (j == 2 ? 0 : 0 - P1));
    uint64_t a[8];
    if (blocks) for (int k = 0; k < 8; k++) __builtin_memcpy(&a[k], p + 8u * ((uint32_t)k * 64u + lane), 8);
    for (uint32_t b = 0; b < blocks; b++) {
        uint64_t (&cur)[512] = prod[b & 1];
        for (int k = 0; k < 8; k++) cur[(uint32_t)k * 64u + lane] = a[k] * P2;
        if (b + 1 < blocks) for (int k = 0; k < 8; k++) __builtin_memcpy(&a[k], p + 4096u * (b + 1) + 8u * ((uint32_t)k * 64u + lane), 8);   // in flight during the chain
        __syncthreads();
        if (lane < 4) {
            for (uint32_t s = 0; s < 128; s += 8) {
                uint64_t m[8];
                for (int q = 0; q < 8; q++) m[q] = cur[4u * (s + (uint32_t)q) + j];
                for (int q = 0; q < 8; q++) v = xxh_rotl(v + m[q], 31) * P1;
            }
        }
    }
    for (uint32_t s = blocks << 7; s < stripes; s++) { uint64_t x; __builtin_memcpy(&x, p + 32u * s + 8u * j, 8); v = xxh_round(v, x); }
    uint64_t vv[4];
    for (int q = 0; q < 4; q++) {
        uint32_t const lo = __shfl((uint32_t)v, q), hi = __shfl((uint32_t)(v >> 32), q);
        vv[q] = ((uint64_t)hi << 32) | lo;
    }
    if (lane == 0) {
        uint64_t h;
        if (n >= 32) {
            h = xxh_rotl(vv[0], 1) + xxh_rotl(vv[1], 7) + xxh_rotl(vv[2], 12) + xxh_rotl(vv[3], 18);
            for (int q = 0; q < 4; q++) h = (h ^ xxh_round(0, vv[q])) * P1 + P4;
        } else h = P5;
        h += (uint64_t)n;
        uint32_t pos = stripes << 5;
        while (pos + 8 <= n) { uint64_t x; __builtin_memcpy(&x, p + pos, 8); h ^= xxh_round(0, x); h = xxh_rotl(h, 27) * P1 + P4; pos += 8; }
        if (pos + 4 <= n) { uint32_t x; __builtin_memcpy(&x, p + pos, 4); h ^= (uint64_t)x * P1; h = xxh_rotl(h, 23) * P2 + P3; pos += 4; }
        while (pos < n) { h ^= (uint64_t)p[pos++] * P5; h = xxh_rotl(h, 11) * P1; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
        checks[ui] = (uint32_t)h;
    }
}

// Stage 3: pack the per-unit slots into one contiguous stream.  offsets[] = exclusive prefix sum of outSize[].
__global__ void __launch_bounds__(256)
k_gather(const uint8_t* __restrict__ outArena, const ZhipSlot* __restrict__ slots, const uint32_t* __restrict__ outSize,
         const uint64_t* __restrict__ offsets, uint32_t nUnits, uint8_t* __restrict__ dst)
{
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    const uint8_t* s = outArena + slots[ui].outOff;
    uint8_t* d = dst + offsets[ui];
    uint32_t const n = outSize[ui];
    // destination alignment is arbitrary: peel to 16 bytes, then 16-byte vectors (source slots are 16-byte aligned)
    uint32_t const head = (uint32_t)((16 - ((uintptr_t)d & 15)) & 15) < n ? (uint32_t)((16 - ((uintptr_t)d & 15)) & 15) : n;
    for (uint32_t i = threadIdx.x; i < head; i += blockDim.x) d[i] = s[i];
    uint32_t const vecs = (n - head) >> 4;
    for (uint32_t i = threadIdx.x; i < vecs; i += blockDim.x) {
        uint4 v; __builtin_memcpy(&v, s + head + 16 * (size_t)i, 16);
        *(uint4*)(d + head + 16 * (size_t)i) = v;
    }
    for (uint32_t i = head + 16 * vecs + threadIdx.x; i < n; i += blockDim.x) d[i] = s[i];
}

// exclusive prefix sum of outSize[0..nUnits) into offsets[0..nUnits] (single workgroup; nUnits is small)
__global__ void __launch_bounds__(256)
k_offsets(const uint32_t* __restrict__ outSize, uint32_t nUnits, uint64_t* __restrict__ offsets)
{
    __shared__ unsigned long long part[256];
    uint32_t const t = threadIdx.x;
    uint32_t const per = (nUnits + 255) / 256;
    uint32_t const a = t * per < nUnits ? t * per : nUnits, b = a + per < nUnits ? a + per : nUnits;
    unsigned long long s = 0;
    for (uint32_t i = a; i < b; i++) s += outSize[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) { unsigned long long acc = 0; for (int i = 0; i < 256; i++) { unsigned long long const v = part[i]; part[i] = acc; acc += v; } offsets[nUnits] = acc; }
    __syncthreads();
    unsigned long long run = part[t];
    for (uint32_t i = a; i < b; i++) { offsets[i] = run; run += outSize[i]; }
}

// The same prefix sum for MANY units (the records workload: 10 M frames): tiles of ZHIP_SCAN_TILE sizes, one workgroup each.
// k_offsets_tiles sums every tile, k_offsets (above) scans the tile sums, k_offsets_apply scans inside each tile from its base.
#define ZHIP_SCAN_TILE 4096u
__global__ void __launch_bounds__(256)
k_offsets_tiles(const uint32_t* __restrict__ outSize, uint32_t nUnits, uint32_t* __restrict__ tileSums)
{
    __shared__ unsigned long long red[4];
    uint32_t const t = threadIdx.x, base = blockIdx.x * ZHIP_SCAN_TILE;
    unsigned long long s = 0;
    for (uint32_t i = base + t; i < base + ZHIP_SCAN_TILE && i < nUnits; i += 256) s += outSize[i];
    for (int d = 32; d; d >>= 1) s += __shfl_down(s, d);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) tileSums[blockIdx.x] = (uint32_t)(red[0] + red[1] + red[2] + red[3]);      // a tile of 4 096 frames of <= 128 KB + header fits 32 bits
}
__global__ void __launch_bounds__(256)
k_offsets_apply(const uint32_t* __restrict__ outSize, uint32_t nUnits, const uint64_t* __restrict__ tileOffs, uint32_t nTiles, uint64_t* __restrict__ offsets)
{
    __shared__ unsigned long long part[256];
    uint32_t const t = threadIdx.x, base = blockIdx.x * ZHIP_SCAN_TILE;
    uint32_t const per = ZHIP_SCAN_TILE / 256;
    uint32_t const a = base + t * per;
    unsigned long long s = 0;
    for (uint32_t i = a; i < a + per && i < nUnits; i++) s += outSize[i];
    part[t] = s;
    __syncthreads();
    if (t == 0) { unsigned long long acc = tileOffs[blockIdx.x]; for (int i = 0; i < 256; i++) { unsigned long long const v = part[i]; part[i] = acc; acc += v; } }
    __syncthreads();
    unsigned long long run = part[t];
    for (uint32_t i = a; i < a + per && i < nUnits; i++) { offsets[i] = run; run += outSize[i]; }
    if (blockIdx.x == 0 && t == 0) offsets[nUnits] = tileOffs[nTiles];
}

// Decoder: persistent 128-thread workgroups, each takes frames from a queue (counter) until it is empty; per workgroup a
// literal buffer and two hand-over buffers of sequence records in HBM/L2.  Dynamic LDS = sizeof(DecShared).
#ifndef ZHIP_DEC_WAVES_PER_EU
#define ZHIP_DEC_WAVES_PER_EU 3          /* 6 workgroups per CU = what the 25 KB of LDS per workgroup allow; keeps the register allocator at <= 168 VGPRs */
#endif
__global__ void __launch_bounds__(ZHIP_DEC_THREADS) __attribute__((amdgpu_waves_per_eu(ZHIP_DEC_WAVES_PER_EU, ZHIP_DEC_WAVES_PER_EU)))
k_decode(const uint8_t* __restrict__ src, const ZhipDFrame* __restrict__ frames, uint32_t nFrames, uint8_t* __restrict__ dst,
         uint8_t* __restrict__ litArena, ZhipDSeq* __restrict__ recArena, uint32_t* __restrict__ counter,
         ZhipDDictDev dict, const uint64_t* __restrict__ defTabs, ZhipDResult* __restrict__ results)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    DecShared* const S = (DecShared*)smem;
    uint8_t* const litBuf = litArena + (size_t)blockIdx.x * ZHIP_DEC_LIT_STRIDE;
    ZhipDSeq* const recBuf = recArena + (size_t)blockIdx.x * 2 * (ZHIP_DEC_CHUNK + 1);
    if (threadIdx.x == 0) { S->dictHufIn = 0; S->dictFseIn = 0; }
    for (;;) {
        if (threadIdx.x == 0) S->frame = atomicAdd(counter, 1u);
        __syncthreads();
        uint32_t const f = S->frame;
        __syncthreads();
        if (f >= nFrames) break;
        ZhipDFrame const fr = frames[f];
        decode_frame(S, src + fr.srcOff, fr.srcLen, dst + fr.dstOff, fr.dstCap, litBuf, recBuf, dict.content ? &dict : nullptr, defTabs, results + f);
    }
}

// ONE large frame, block-parallel (zhip_decode_big.h): src = the frame, out = its content; the launches in order
__global__ void __launch_bounds__(64)
k_bf_walk(const uint8_t* __restrict__ src, uint32_t srcLen, uint32_t hdrSize, uint32_t blockMax, uint32_t hasChecksum,
          ZhipBfBlock* __restrict__ blocks, uint32_t capBlocks, ZhipBfInfo* __restrict__ info)
{
    bf_walk(src, srcLen, hdrSize, blockMax, hasChecksum, blocks, capBlocks, info);
}
__global__ void __launch_bounds__(256)
k_bf_prep(const uint8_t* __restrict__ src, uint32_t blockMax, ZhipBfBlock* __restrict__ blocks, const ZhipBfInfo* __restrict__ info)
{
    uint32_t const bi = blockIdx.x * 256 + threadIdx.x;
    if (bi < info->nBlocks) bf_prep(src, blockMax, blocks, bi);
}
__global__ void __launch_bounds__(64)
k_bf_deps(ZhipBfBlock* __restrict__ blocks, ZhipBfInfo* __restrict__ info) { bf_deps(blocks, info); }
// dynamic LDS = sizeof(DecShared); grid = number of blocks
__global__ void __launch_bounds__(ZHIP_BF_THREADS)
k_bf_entropy(const uint8_t* __restrict__ src, uint32_t blockMax, ZhipBfBlock* __restrict__ blocks, const ZhipBfInfo* __restrict__ info,
             uint8_t* __restrict__ litArena, ZhipDSeq* __restrict__ recArena, const uint64_t* __restrict__ defTabs)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    if (blockIdx.x < info->nBlocks) bf_entropy_block((DecShared*)smem, src, blockMax, blocks, blockIdx.x, litArena, recArena, defTabs);
}
__global__ void __launch_bounds__(64)
k_bf_scan(ZhipBfBlock* __restrict__ blocks, ZhipBfInfo* __restrict__ info, uint32_t dstCap)
{
    __shared__ uint32_t sh[64 * 3];
    bf_scan(blocks, info, dstCap, sh);
}
__global__ void __launch_bounds__(256)
k_bf_build(const uint8_t* __restrict__ src, const ZhipBfBlock* __restrict__ blocks, ZhipBfInfo* __restrict__ info, const uint8_t* __restrict__ litArena,
           const ZhipDSeq* __restrict__ recArena, uint8_t* __restrict__ out, uint32_t* __restrict__ map)
{
    if (blockIdx.x < info->nBlocks && info->status == 0) bf_build_block(src, blocks, blockIdx.x, litArena, recArena, out, map, info);
}
__global__ void __launch_bounds__(256)
k_bf_jump(uint32_t* __restrict__ map, uint32_t n, ZhipBfInfo* __restrict__ info) { bf_jump(map, n, &info->changed); }
__global__ void __launch_bounds__(256)
k_bf_copy(const uint32_t* __restrict__ map, uint8_t* __restrict__ out, uint32_t n) { bf_copy(map, out, n); }

// content checksums: checks[] = XXH64 low words of the decoded frames (k_xxh64 over the destination)
__global__ void __launch_bounds__(256)
k_dec_verify(ZhipDResult* __restrict__ results, const uint32_t* __restrict__ checks, uint32_t nFrames)
{
    uint32_t const i = blockIdx.x * 256 + threadIdx.x;
    if (i < nFrames && results[i].status == 0 && results[i].hasChecksum && results[i].checksum != checks[i]) { results[i].status = ZHIP_DE_CHECKSUM; results[i].size = 0; }
}

// ---- stage-test hooks (tests/test_emu_tables.py, tests/test_gpu_tables.py): the wave-wide table builders of zhip_tables.h
// on caller-supplied histograms, one 64-thread workgroup per case, so that each stage is pinned to the reference's own stage
// function (HUF_buildCTable_wksp / HUF_writeCTable_wksp, FSE_normalizeCount / FSE_writeNCount / FSE_buildCTable_wksp).
struct ZhipTestHufShared { HufWork w; uint32_t count[256]; uint32_t code[256]; uint8_t hdr[136]; };
__global__ void __launch_bounds__(64)
k_test_huf(const uint32_t* __restrict__ counts /* nCases x 256 */, const uint32_t* __restrict__ maxSyms, uint32_t maxNbBits,
           uint32_t* __restrict__ codes /* nCases x 256 */, uint8_t* __restrict__ hdrs /* nCases x 136 */, uint32_t* __restrict__ meta /* nCases x 2: table log, header size */)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    ZhipTestHufShared* const S = (ZhipTestHufShared*)smem;
    uint32_t const c = blockIdx.x, lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) S->count[i] = counts[(size_t)c * 256 + i];
    __builtin_amdgcn_wave_barrier();
    uint32_t const log = huf_build_codes_wave(&S->w, S->count, maxSyms[c], maxNbBits, S->code);
    uint32_t const h = huf_write_table_wave(&S->w, S->hdr, S->code, maxSyms[c], log);
    for (uint32_t i = lane; i < 256; i += 64) codes[(size_t)c * 256 + i] = S->code[i];
    for (uint32_t i = lane; i < 136; i += 64) hdrs[(size_t)c * 136 + i] = i < h ? S->hdr[i] : 0;
    if (lane == 0) { meta[2 * c] = log; meta[2 * c + 1] = h; }
}
struct ZhipTestFseShared { FseCTable ct; uint32_t count[64]; int16_t norm[64]; uint32_t words[16]; uint8_t ncount[64]; uint8_t cellSym[4096]; uint16_t first[64]; };
__global__ void __launch_bounds__(64)
k_test_fse(const uint32_t* __restrict__ counts /* nCases x 64 */, const uint32_t* __restrict__ params /* nCases x 4: total, maxSym, tableLog, useLowProb */,
           int16_t* __restrict__ norms /* nCases x 64 */, uint8_t* __restrict__ ncounts /* nCases x 64 */, int32_t* __restrict__ meta /* nCases x 2: normalize rc, NCount size */,
           FseCTable* __restrict__ tables)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    ZhipTestFseShared* const S = (ZhipTestFseShared*)smem;
    uint32_t const c = blockIdx.x, lane = threadIdx.x;
    uint32_t const total = params[4 * c], maxSym = params[4 * c + 1], tableLog = params[4 * c + 2], lowProb = params[4 * c + 3];
    S->count[lane] = counts[(size_t)c * 64 + lane]; S->norm[lane] = 0; S->ncount[lane] = 0;
    {   uint32_t* const z = (uint32_t*)&S->ct; for (uint32_t i = lane; i < sizeof(FseCTable) / 4; i += 64) z[i] = 0; }
    __builtin_amdgcn_wave_barrier();
    int const rc = fse_normalize_wave(S->norm, tableLog, S->count, total, maxSym, lowProb != 0);
    uint32_t sz = 0;
    if (rc == 1) {
        sz = fse_write_ncount_wave(S->words, S->ncount, S->norm, maxSym, tableLog);
        if (sz) fse_build_ctable_wave(&S->ct, S->norm, maxSym, tableLog, S->cellSym, S->first);
    }
    norms[(size_t)c * 64 + lane] = S->norm[lane];
    ncounts[(size_t)c * 64 + lane] = lane < sz ? S->ncount[lane] : 0;
    if (lane == 0) { meta[2 * c] = rc; meta[2 * c + 1] = (int32_t)sz; }
    {   const uint32_t* const f = (const uint32_t*)&S->ct; uint32_t* const t = (uint32_t*)&tables[c]; for (uint32_t i = lane; i < sizeof(FseCTable) / 4; i += 64) t[i] = f[i]; }
}

}  // namespace zhip
