// zhip_kernels_entropy.h — __global__ entry points: stage 2 (literals + sequences entropy coding, frame assembly), checksums, gather / prefix sums, stage-test hooks.
// Compiled into its own code object by zhip_k_entropy.hip: a change in another kernel family cannot move this one's inlining or register allocation
// (round 3 ended on a decoder whose code the block-parallel decoder's arrival had reshaped).  Declarations for the host side: zhip_kernel_decls.h.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_kernel_params.h"
#include "zhip_entropy.h"

// register caps for more resident wavefronts (A/B-measured): empty = the compiler's own choice
#ifndef ZHIP_DFAST_OCC
#define ZHIP_DFAST_OCC __attribute__((amdgpu_waves_per_eu(4)))   /* with the window (131 VGPRs as compiled): 4 waves per SIMD, A/B on 2 GiB: 3 / 4 / 5 / 6 -> text 195 / 169 / 188 / 252 ms */
#endif
#ifndef ZHIP_LAZY_OCC
#define ZHIP_LAZY_OCC
#endif
#ifndef ZHIP_ENT_OCC
#define ZHIP_ENT_OCC __attribute__((amdgpu_waves_per_eu(8)))    /* round 6, after the stage's LDS shrank to 19.3 KB (eight workgroups per CU): 64 registers = eight wavefronts per SIMD: datagen 1.08 -> 0.96 ms per GiB, Silesia-shaped 2.04 -> 1.89, text even (profiles/r06_ab_entropy_occupancy8.log; a wash on round 5's kernel) */
#endif

namespace zhip {

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

// Plugin boundary (zhip_prepare_sequences): the blocks' sequence records, each in its own slot of the sequence arena, packed back to back
// (offs[] = exclusive prefix sum of metas[].nbSeq) so that ONE copy brings them to the host
__global__ void __launch_bounds__(256)
k_seq_compact(const ZhipSeq* __restrict__ seqs, const ZhipSlot* __restrict__ slots, const ZhipParse* __restrict__ metas,
              const uint64_t* __restrict__ offs, uint32_t nUnits, ZhipSeq* __restrict__ dst)
{
    uint32_t const ui = blockIdx.x;
    if (ui >= nUnits) return;
    const ZhipSeq* s = seqs + slots[ui].seqOff;
    ZhipSeq* d = dst + offs[ui];
    uint32_t const ns = metas[ui].nbSeq;
    for (uint32_t i = threadIdx.x; i < ns; i += 256) d[i] = s[i];
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

// ---- stage-test hooks (tests/test_emu_tables.py, tests/test_gpu_tables.py): the wave-wide table builders of zhip_tables.h
// on caller-supplied histograms, one 64-thread workgroup per case, so that each stage is pinned to the reference's own stage
// function (HUF_buildCTable_wksp / HUF_writeCTable_wksp, FSE_normalizeCount / FSE_writeNCount / FSE_buildCTable_wksp).
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
