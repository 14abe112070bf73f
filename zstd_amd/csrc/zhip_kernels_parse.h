// zhip_kernels_parse.h — __global__ entry points: stage 1 of the unit path: ZSTD_fast / ZSTD_dfast match finders (plain, ticket-queue, attached-dictionary and copied-dictionary forms) and the dispatch order.
// Compiled into its own code object by zhip_k_parse.hip: a change in another kernel family cannot move this one's inlining or register allocation
// (round 3 ended on a decoder whose code the block-parallel decoder's arrival had reshaped).  Declarations for the host side: zhip_kernel_decls.h.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_kernel_params.h"
#include "zhip_parse.h"
#include "zhip_parse_dfast.h"
#include "zhip_parse_dict.h"
#include "zhip_parse_ext.h"

// register caps for more resident wavefronts (A/B-measured): empty = the compiler's own choice
#ifndef ZHIP_DICT_OCC
#define ZHIP_DICT_OCC __attribute__((amdgpu_waves_per_eu(7)))   /* the dictionary finders' queue kernels had grown to 87 / 88 VGPRs = five wavefronts per SIMD; at 72 = seven: 10 M-record stage -7 % (6: -5.6 %, 8: spills, even) — profiles/r06_ab_records_parser_occupancy.log */
#endif
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
#define ZHIP_FAST_OCC __attribute__((amdgpu_waves_per_eu(4)))      /* 128 VGPRs (round 6: the window's state was cut to fit them — repcode state as two XOR words instead of four masks, the sequences stored as they are found,
                                                                      no deferred literal chunk, the schedule's lane offsets recomputed — 130-135 without the cap, 48 bytes of scratch with it): four wavefronts per SIMD = sixteen per CU,
                                                                      eight on LDS tables and up to eight on global ones.  Round 5's 152 registers held three. */
#endif
__global__ void __launch_bounds__(64) ZHIP_FAST_OCC
k_parse_fast(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, uint32_t nUnits,
             ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    ZhipUnit const u = units[ui];
    if (u.strategy != ZHIP_STRAT_FAST) return;          // another family's kernel handles it
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
// nullptr = as they come).  Two kernels share ONE queue: k_parse_fast_q keeps its table in LDS (eight wavefronts fill a CU's LDS),
// k_parse_fast_g keeps it in global memory and needs no LDS at all, so its wavefronts run BESIDE the nine on the same CU and hide the
// latency those cannot.  Whoever is free takes the next unit: the split between the two adjusts itself.
__device__ __forceinline__ uint32_t queue_take(uint32_t* queue)
{
    uint32_t t = 0;
    if ((threadIdx.x & 63) == 0) t = atomicAdd(queue, 1u);
    return __builtin_amdgcn_readfirstlane(t);
}
// Issue priority by rank (round 6).  With the heaviest units first, a batch that is only one or two rounds of the resident wavefronts deep takes as long as its
// heaviest unit takes on a CU it shares with fifteen others (Silesia-shaped x4: 6 468 units on 4 096 slots — the 128-register parser gained nothing there although
// 8 192 units of the same mix gained 18 %).  The first tickets therefore run at a higher s_setprio: about one wavefront per SIMD at 3, one more at 2, two more at 1.
#ifndef ZHIP_FAST_PRIO
#define ZHIP_FAST_PRIO 1
#endif
__device__ __forceinline__ void queue_prio(uint32_t t, bool ranked)
{
    if (!ZHIP_FAST_PRIO) return;
    if (!ranked || t >= 4096u) __builtin_amdgcn_s_setprio(0);
    else if (t < 1024u) __builtin_amdgcn_s_setprio(3);
    else if (t < 2048u) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);
}
#ifndef ZHIP_FASTG_OCC
#define ZHIP_FASTG_OCC __attribute__((amdgpu_waves_per_eu(4)))
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
        if (u.strategy != ZHIP_STRAT_FAST) continue;
        queue_prio(t, order != nullptr);
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
    {   uint32_t const lim = __builtin_amdgcn_readfirstlane(queue[1]);       // k_order_sort's decision: the workgroups from `lim` on stay out (0: no limit)
        if (lim && blockIdx.x >= lim) return; }
    for (;;) {
        uint32_t const t = queue_take(queue);
        if (t >= nUnits) return;
        uint32_t const ui = order ? order[t] : t;
        ZhipUnit const u = units[ui];
        if (u.strategy != ZHIP_STRAT_FAST) continue;
        queue_prio(t, order != nullptr);
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
// It also decides how many global-table wavefronts join the queue (round 6): `gLimit` (the queue's second word, read by k_parse_fast_g; 0 = all that
// were launched) is set to `gSparse` when the batch's mean estimated cost is below `denseCost` — a batch with few sequences per unit spends its time in
// table gathers, and more than six global tables per CU then miss the L2 (datagen: +8 % with eight), a dense one spends it in the event loop and takes
// all eight (Silesia-shaped -5 %, text -4 % against six; profiles/r06_ab_fast_128_registers.log).  nullptr: no decision.
__global__ void __launch_bounds__(1024)
k_order_sort(const uint32_t* __restrict__ cost, uint32_t nUnits, uint32_t* __restrict__ order, uint32_t* __restrict__ gLimit, uint32_t gSparse, uint32_t denseCost)
{
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t part[1024];
    __shared__ unsigned long long costSum;
    uint32_t const tid = threadIdx.x;
    hist[tid] = 0; hist[tid + 1024] = 0;
    if (tid == 0) costSum = 0;
    __syncthreads();
    unsigned long long mine = 0;
    for (uint32_t i = tid; i < nUnits; i += 1024) { uint32_t const cst = cost[i], b = cst >> 4; mine += cst; atomicAdd(&hist[2047u - (b < 2047u ? b : 2047u)], 1u); }
    if (gLimit) atomicAdd(&costSum, mine);
    __syncthreads();
    if (gLimit && tid == 0 && costSum < (unsigned long long)denseCost * nUnits) *gLimit = gSparse;
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
              ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, ZhipParse* __restrict__ metas, uint32_t* __restrict__ queue)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    // persistent workgroups on a ticket counter (queue; nullptr: workgroup w takes the units w, w + gridDim.x, ...): every workgroup reuses ONE
    // table pair (tabs + w * tabStride) for all its units, so the table memory in use is gridDim.x pairs — what is resident — not nUnits pairs
    // (Silesia-shaped x64: 1.5 GB of tables instead of 40 GB)
    for (uint32_t ui = queue ? queue_take(queue) : blockIdx.x; ui < nUnits; ui = queue ? queue_take(queue) : ui + gridDim.x) {
        ZhipUnit const u = units[ui];
        if (u.strategy != ZHIP_STRAT_DFAST) continue;
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
__global__ void __launch_bounds__(64) ZHIP_DICT_OCC
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
__global__ void __launch_bounds__(64) ZHIP_DICT_OCC
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

}  // namespace zhip
