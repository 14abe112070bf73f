// zhip_kernels_frames.h — __global__ entry points: multi-block frames and job-pool frames (fast / dfast: k_frame_*; lazy strategies: k_lz_* + k_frame_lazy).
// Compiled into its own code object by zhip_k_frames.hip: a change in another kernel family cannot move this one's inlining or register allocation
// (round 3 ended on a decoder whose code the block-parallel decoder's arrival had reshaped).  Declarations for the host side: zhip_kernel_decls.h.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_kernel_params.h"
#include "zhip_parse.h"
#include "zhip_parse_dfast.h"
#include "zhip_parse_lazy.h"
#include "zhip_entropy.h"
#include "zhip_frame.h"
#include "zhip_frame_lazy.h"

namespace zhip {

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
    // (the four-per-CU variant keeps every table in HBM, also the ones that would fit LDS: the host launches it for batches of more workgroups than the LDS form holds at once)
    uint32_t const mode = OCC == 4 ? (uint32_t)ZHIP_FT_HBM : frame_table_mode(u.strategy, u.hashLog, (uint64_t)u.srcLen + (job ? job->prefixLen + 1u : 0u));
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
            const uint32_t* __restrict__ prev, const uint8_t* __restrict__ tags, LzRec* __restrict__ best,
            const ZhipFrameState* __restrict__ states /* the pass after k_lz_predict: only the windows it marked; nullptr: all */)
{
    uint32_t const wi = wBase + blockIdx.y;
    if (wi >= nW) return;
    if (states && !states[wi].predicted) return;
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
             uint32_t* __restrict__ prev, const uint8_t* __restrict__ tags, const LzRec* __restrict__ best, ZhipFrameState* __restrict__ states)
{
    HIP_DYNAMIC_SHARED(unsigned char, smem)
    uint32_t const wi = blockIdx.x;
    if (wi >= nW) return;
    ZhipUnit const u = units[wi];
    bool marked = false;
    if (u.strategy >= ZHIP_STRAT_GREEDY && u.srcLen != 0) {
        ZhipLzSlot const L = lz[wi];
        marked = frame_lazy_predict(lz_window(src, u, jobs, wi), u, prev + L.posOff, tags + L.posOff, best + L.posOff, (ZhipParse*)smem, jobs ? jobs + wi : (const ZhipJob*)nullptr);
    }
    if (threadIdx.x == 0) states[wi].predicted = marked ? 1u : 0u;           // k_frame_lazy compares its decisions with the marks only where there are any
}
// k_frame_lazy: dynamic LDS = frame_lazy_lds_bytes()
__global__ void __launch_bounds__(ZHIP_ENT_THREADS, 2)
k_frame_lazy(const uint8_t* __restrict__ src, const ZhipUnit* __restrict__ units, const ZhipSlot* __restrict__ slots, const ZhipJob* __restrict__ jobs,
             const ZhipLzSlot* __restrict__ lz, uint32_t nW, uint32_t* __restrict__ prev, const uint8_t* __restrict__ tags, const LzRec* __restrict__ best,
             uint32_t* __restrict__ heads, uint8_t* __restrict__ rings /* the live rows' arena, or nullptr */, ZhipSeq* __restrict__ seqs, uint8_t* __restrict__ lits, uint16_t* __restrict__ stBits,
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
    frame_lazy(lz_window(src, u, jobs, wi), u, L, prev + L.posOff, tags + L.posOff, best + L.posOff, heads + L.headOff, rings ? rings + L.ringOff : (uint8_t*)nullptr,
               seqs + sl.seqOff, lits + sl.litOff, stBits + 3 * sl.seqOff, sl.seqCap, out + sl.outOff, outSize + wi, sh, fs, states + wi, ck, cv, job, havePred != 0);
}

}  // namespace zhip
