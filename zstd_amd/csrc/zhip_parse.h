// zhip_parse.h — gfx950 match finder for strategy ZSTD_fast, one wavefront per 128 KB unit.
//
// WHAT it computes: exactly the sequences the reference's ZSTD_compressBlock_fast_noDict_generic
// (lib/compress/zstd_fast.c:192-423) emits for a unit with no history (fresh table, rep = {1,4,8}).
//
// HOW (CDNA4 design, not a translation).  The reference walks positions one or two at a time because each lookup
// sees the table writes of the positions before it.  Here one 64-lane wavefront owns the unit and evaluates a
// *batch* of 32 reference iterations (64 search positions) at once:
//   * the positions the reference would visit from the current point are a data-independent schedule (pairs
//     A_k, A_k+1 with a gap that grows every 128 bytes, zstd_fast.c:232-347); lane 2k+b takes A_k+b, even lanes also
//     carry the repcode probe of iteration k (at A_{k+1});
//   * every lane hashes its position and gathers the table entry from LDS.  The table is wave-private LDS:
//     16-bit entries + a 1-bit plane for bit 16 of the position = 17 KB for hashLog 13, so NINE units are resident
//     per CU (LDS is allocated in 1280-byte granules on gfx950: 9 x 14 granules);
//   * lanes of one batch that hash alike must see each other's inserts in lane order.  A 512-byte LDS scratch
//     (write lane id / read back) finds the colliding lanes, ballots turn them into exact per-hash lane groups, and
//     each lane takes the position of its closest earlier group member as its candidate — so ONE pass is exact;
//   * ballots give the first event in the reference's own order (repcode at ip2, match at ip0, match at ip1);
//     the lanes the reference would have inserted before that event write the table (last lane of a group only);
//   * the unit is latency-bound (dependent global loads), so the code is organised around global round trips:
//     per batch ONE (candidate bytes; the next batch's source bytes are loaded speculatively in its shadow), per
//     match TWO: forward+backward extension in one wave-wide compare (48 x 8 B forward, 16 x 8 B backward), then one
//     round that fetches the bytes for the two complementary inserts, the immediate-repcode probe+count and the
//     next batch.
// All control flow is wave-uniform (derived from ballots); LDS traffic is wave-private, so no s_barrier is needed.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"

#ifndef ZHIP_LDS
#define ZHIP_LDS __attribute__((address_space(3)))          /* the host SIMT emulator (tests/simt) defines it empty */
#endif

namespace zhip {

typedef ZHIP_LDS uint8_t  lds_u8;
typedef ZHIP_LDS uint16_t lds_u16;
typedef ZHIP_LDS uint32_t lds_u32;

// ------------------------------------------------------------------ unaligned source access (HBM through L1/L2)
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }   // -> SGPR
__device__ __forceinline__ int first_lane(unsigned long long m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l)
{
    return (uint64_t)__builtin_amdgcn_readlane((uint32_t)v, l) | ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(v >> 32), l) << 32);
}
__device__ __forceinline__ unsigned long long below_mask(int l) { return l >= 64 ? ~0ull : (l <= 0 ? 0ull : ((1ull << l) - 1)); }

// multiplicative hashes of lib/compress/zstd_compress_internal.h:820-862, evaluated with 32-bit multiplies:
// only the top hBits of the low 64 bits of the product are needed, and (bytes << s) * prime == bytes * (prime << s)
// modulo 2^64, so the shift is folded into the constant.
__device__ __forceinline__ uint32_t mulhi64_top32(uint64_t x, uint64_t p)
{   // high 32 bits of the low 64 bits of x*p
    uint32_t const xl = (uint32_t)x, xh = (uint32_t)(x >> 32), pl = (uint32_t)p, ph = (uint32_t)(p >> 32);
    return __umulhi(xl, pl) + xh * pl + xl * ph;
}
template <uint32_t MLS>
__device__ __forceinline__ uint32_t hash_pos(uint64_t bytes, uint32_t hshift /* 32 - hashLog */)
{
    if (MLS <= 4) return ((uint32_t)bytes * 2654435761U) >> hshift;
    if (MLS == 5) return mulhi64_top32(bytes, 889523592379ULL << 24) >> hshift;
    if (MLS == 6) return mulhi64_top32(bytes, 227718039650203ULL << 16) >> hshift;
    if (MLS == 7) return mulhi64_top32(bytes, 58295818150454627ULL << 8) >> hshift;
    return mulhi64_top32(bytes, 0xCF1BBCDCB7A56463ULL) >> hshift;
}

// ------------------------------------------------------------------ the wave-private hash table in LDS
// value = position in the unit, 0 = empty (position 0 is never inserted, zstd_fast.c:238).  Positions are < 2^17:
// lo[] holds bits 0..15, one bit per entry in hi[] holds bit 16.  Positions are inserted in increasing order, so
// once a position >= 65536 exists every later insert sets its hi bit: the plane only ever needs OR, and it only
// needs to be read once the scan has passed 64 KB.
struct FastTab {
    lds_u16* lo;
    lds_u32* hi;
};
__host__ __device__ inline uint32_t fast_hi_bytes(uint32_t hlog) { uint32_t const b = (1u << hlog) >> 3; return b < 4 ? 4 : b; }
__host__ __device__ inline uint32_t fast_lds_bytes(uint32_t hlog) { return (2u << hlog) + fast_hi_bytes(hlog); }

__device__ __forceinline__ void tab_put(const FastTab& T, uint32_t h, uint32_t pos)
{
    T.lo[h] = (uint16_t)pos;
    if (pos >> 16) __hip_atomic_fetch_or(&T.hi[h >> 5], 1u << (h & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// ------------------------------------------------------------------ wave-wide match extension
// Every load below is clamped to [0, n-8] so that no lane ever reads outside the unit; `sh` bytes are then shifted out.
// equal leading bytes (0..8) of the 8-byte windows at q and q-off, bounded by the end of the unit (nm8 = n - 8)
__device__ __forceinline__ uint32_t lane_same_fwd(const uint8_t* src, uint32_t q, uint32_t off, uint32_t nm8)
{
    uint32_t const qc = q < nm8 ? q : nm8, sh = q - qc;
    uint64_t x = ld64(src + qc) ^ ld64(src + (qc - off));
    x >>= 8 * (sh & 7);
    uint32_t const lim = 8 - sh;                                           // bytes of the window inside the unit
    uint32_t const same = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : lim;
    return sh >= 8 ? 0 : same;
}
// Common-prefix length of src[a..) and src[b..) (b < a) — ZSTD_count (zstd_compress_internal.h:771), 512 B per round
__device__ __forceinline__ uint32_t wave_count_fwd(const uint8_t* src, uint32_t a, uint32_t b, uint32_t nm8)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t total = 0;
    for (;;) {
        uint32_t const same = lane_same_fwd(src, a + 8u * lane, a - b, nm8);
        unsigned long long const stop = __ballot(same < 8);
        if (stop) {
            int const f = first_lane(stop);
            return total + 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f);
        }
        total += 512; a += 512; b += 512;
    }
}

// Number of equal bytes walking backwards from src[ip-1] / src[m-1], at most `limit` (zstd_fast.c:387-391).
__device__ __forceinline__ uint32_t wave_count_back(const uint8_t* src, uint32_t ip, uint32_t m, uint32_t limit)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t total = 0;
    for (;;) {
        uint32_t const i = total + lane;
        bool const stopHere = (i >= limit) || (src[ip - 1 - i] != src[m - 1 - i]);
        unsigned long long const stop = __ballot(stopHere);
        if (stop) return total + (uint32_t)first_lane(stop);
        total += 64;
    }
}

// Backward (at most `lim` bytes before mpos / cand) and forward (from mpos+4 / cand+4) extension of a 4-byte match in
// ONE round of loads: lanes 0..47 compare 8 bytes forward each, lanes 48..63 8 bytes backward each.
__device__ __forceinline__ void wave_extend(const uint8_t* src, uint32_t nm8, uint32_t mpos, uint32_t cand, uint32_t lim,
                                            uint32_t& backLen, uint32_t& fwdLen)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const off = mpos - cand;
    bool const fwd = lane < 48;
    // forward lane: window at q = mpos+4+8*lane.  backward lane j: the r (<= 8) bytes that end at mpos-8j
    uint32_t const q = mpos + 4 + 8u * lane;
    uint32_t const j8 = 8u * (lane - 48);
    uint32_t const rr = lim - j8;
    uint32_t const r = (fwd || lim <= j8) ? 0 : (rr < 8 ? rr : 8);
    uint32_t const qb = r ? mpos - j8 - r : mpos;
    uint32_t const qf = q < nm8 ? q : nm8;
    uint32_t const qc = fwd ? qf : qb;
    uint64_t const x = ld64(src + qc) ^ ld64(src + (qc - off));
    uint32_t same;
    if (fwd) {
        uint32_t const sh = q - qf;
        uint64_t const y = x >> (8 * (sh & 7));
        uint32_t const s = y ? (uint32_t)(__ffsll((long long)y) - 1) >> 3 : 8 - sh;
        same = sh >= 8 ? 0 : s;
    } else {
        uint64_t const y = x << (8 * ((8 - r) & 7));                       // byte r-1 (closest to mpos) -> top byte
        uint32_t const s = y ? (uint32_t)__clzll((long long)y) >> 3 : r;
        same = r ? s : 0;
    }
    unsigned long long const stop = __ballot(same < 8);
    unsigned long long const stopF = stop & 0x0000FFFFFFFFFFFFull, stopB = stop >> 48;
    if (stopF) { int const f = first_lane(stopF); fwdLen = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f); }
    else fwdLen = 384 + wave_count_fwd(src, mpos + 4 + 384, cand + 4 + 384, nm8);
    if (stopB) { int const f = first_lane(stopB); backLen = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f + 48); }
    else backLen = 128 + wave_count_back(src, mpos - 128, cand - 128, lim - 128);
}

// ------------------------------------------------------------------ the parser
struct FastOut {
    ZhipSeq* seqs;          // global, capacity ZHIP_SEQ_CAP
    uint8_t* lits;          // global, the unit's literal buffer (ZHIP_LIT_STRIDE bytes)
    uint32_t nbSeq, longPos, longType;
    uint32_t litPos;        // literals emitted so far
    uint64_t pendV;         // per lane: 8 loaded literal bytes of the most recent run, stored at the next call
    uint32_t pendSh;        // per lane: bits to shift pendV right by (loads are clamped to the unit)
    uint32_t pendOff, pendLen;
};

__device__ __forceinline__ void st64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }

// Literal copy (the job of ZSTD_storeSeq's wildcopy, zstd_compress_internal.h:684-700), kept off the parser's
// critical path: a run's first 512 bytes are LOADED when the sequence is emitted (they are hot in L1/L2: the
// scan just read them) and STORED at the next call, so nobody waits for the load.  Like the reference's wildcopy the
// last 8-byte chunk of a run may spill up to 7 bytes past it; the next run (stored later) overwrites them, and the
// buffer has slack after the last one.
__device__ __forceinline__ void lits_flush(FastOut& o)
{
    uint32_t const lane8 = 8u * (uint32_t)lane_id();
    if (lane8 < o.pendLen) st64(o.lits + o.pendOff + lane8, o.pendV >> o.pendSh);
    o.pendLen = 0;
    __builtin_amdgcn_wave_barrier();        // later runs overwrite this run's spill: keep the stores in program order
}
__device__ __forceinline__ void lits_copy(FastOut& o, const uint8_t* src, uint32_t nm8, uint32_t from, uint32_t len)
{
    lits_flush(o);
    if (len == 0) return;
    uint32_t const lane8 = 8u * (uint32_t)lane_id();
    {   uint32_t const q = from + lane8, qc = q < nm8 ? q : nm8, sh = q - qc;      // sh <= 7 whenever lane8 < len
        if (lane8 < len) { o.pendV = ld64(src + qc); o.pendSh = 8 * (sh & 7); }          // not consumed here: no wait
        o.pendOff = o.litPos; o.pendLen = len < 512 ? len : 512;
    }
    for (uint32_t off = 512; off < len; off += 512) {
        uint32_t const q = from + off + lane8, qc = q < nm8 ? q : nm8, sh = q - qc;
        if (off + lane8 < len) st64(o.lits + o.litPos + off + lane8, ld64(src + qc) >> (8 * (sh & 7)));
    }
    o.litPos += len;
}

__device__ __forceinline__ void store_seq(FastOut& o, uint32_t litLength, uint32_t offBase, uint32_t matchLength)
{   // zstd_compress_internal.h:671-728 minus the literal copy (literals are gathered by the entropy kernel)
    uint32_t const mlBase = matchLength - 3;
    if (litLength > 0xFFFF) { o.longType = 1; o.longPos = o.nbSeq; }
    if (mlBase > 0xFFFF) { o.longType = 2; o.longPos = o.nbSeq; }
    if (lane_id() == 0) {
        ZhipSeq s; s.offBase = offBase; s.litLength = (uint16_t)litLength; s.mlBase = (uint16_t)mlBase;
        o.seqs[o.nbSeq] = s;
    }
    o.nbSeq++;
}

// source bytes one batch needs, per lane: lane 2k+b searches A_k+b; even lanes also probe the repcode at A_{k+1}
struct FastBatch {
    uint64_t bytes;         // 8 bytes at pos
    uint32_t rcur;          // 4 bytes at rpos = A_{k+1}
    uint32_t rv;            // 4 bytes at rpos - rep1
};
// lane offsets of the schedule from its first position: pos = ip0 + posOff, rpos = ip0 + rposOff
__device__ __forceinline__ void batch_offsets(uint32_t g0, uint32_t step, uint32_t& posOff, uint32_t& rposOff)
{
    uint32_t const lane = (uint32_t)lane_id(), k = lane >> 1;
    posOff = (k ? g0 + (k - 1) * step : 0) + (lane & 1);
    rposOff = g0 + k * step;
}
// loads are clamped to the unit, so a speculative batch beyond the end reads harmless bytes
__device__ __forceinline__ FastBatch batch_load(const uint8_t* src, uint32_t nm8, uint32_t ip0, uint32_t posOff, uint32_t rposOff, uint32_t rep1)
{
    uint32_t const p = ip0 + posOff, r = ip0 + rposOff;
    uint32_t const pc = p < nm8 ? p : nm8, rc = r < nm8 ? r : nm8;
    FastBatch b;
    b.bytes = ld64(src + pc);
    b.rcur = ld32(src + rc);
    b.rv = ld32(src + (rc - rep1));
    return b;
}

// smem: fast_lds_bytes(hashLog) bytes of wave-private LDS
template <uint32_t MLS>
__device__ inline void parse_fast_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u,
                                       unsigned char* smem, ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const hlog = u.hashLog, hshift = 32 - hlog;
    uint32_t const stepSize = u.targetLength + !u.targetLength + 1;         // zstd_fast.c:200
    FastOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.longPos = 0; out.longType = 0;
    out.litPos = 0; out.pendV = 0; out.pendSh = 0; out.pendOff = 0; out.pendLen = 0;
    ZPROF_DECL

    FastTab T;
    T.lo = (lds_u16*)(uintptr_t)smem;
    T.hi = (lds_u32*)(uintptr_t)(smem + (2u << hlog));
    {   // fresh table (zstd_compress.c:2020): lo[] and hi[] are contiguous
        lds_u32* const z = (lds_u32*)(uintptr_t)smem;
        uint32_t const words = fast_lds_bytes(hlog) >> 2;
        for (uint32_t i = lane; i < words; i += 64) z[i] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    ZPROF(0);

    uint32_t anchor = 0, rep1 = 1, rep2 = 4, saved1 = 0, saved2 = 0;
    // :238-244  ip0 = 1, lowest index 0 -> maxRep = 1
    if (rep2 > 1) { saved2 = rep2; rep2 = 0; }
    if (rep1 > 1) { saved1 = rep1; rep1 = 0; }

    if (n >= 13) {                          // shortest unit whose first iteration runs (ip3 = 1 + 2 + 1 < n - 8)
    uint32_t const nm8 = n - 8;
    int32_t const ilimit = (int32_t)nm8;
    uint32_t ip0 = 1;
    uint32_t startPosOff, startRposOff; batch_offsets(stepSize, stepSize, startPosOff, startRposOff);
    unsigned long long const evenLanes = 0x5555555555555555ull;

    bool have = false;                      // `cur` already holds the bytes of the batch that starts at ip0
    FastBatch cur; cur.bytes = 0; cur.rcur = 0; cur.rv = 0;
    for (;;) {                                                               // one turn per `_start`
        uint32_t step = stepSize, g0 = stepSize, nextStep = ip0 + 128;
        if ((int32_t)(ip0 + g0 + 1) >= ilimit) break;                        // :257
        uint32_t posOff = startPosOff, rposOff = startRposOff;
        if (!have) cur = batch_load(src, nm8, ip0, posOff, rposOff, rep1);
        have = false;

        // ---- scan batches until an event or the end of the unit
        int evKind = 0;                      // 0 none (unit finished), 1 match, 2 repcode
        uint32_t mpos = 0, cand0 = 0, cur0 = 0;
        for (;;) {
            // iterations this batch covers: iteration k+1 runs iff A_{k+2}+1 < ilimit (:347); the gap grows after the
            // iteration whose A_{k+2} reaches nextStep (:342-346) — a batch ends there
            uint32_t const pos = ip0 + posOff, rpos = ip0 + rposOff;
            uint32_t const A2 = rpos + step;
            unsigned long long const mEnd = __ballot((int32_t)(A2 + 1) >= ilimit);
            unsigned long long const mInc = __ballot((int32_t)A2 >= (int32_t)nextStep);
            int const kEnd = mEnd ? first_lane(mEnd) >> 1 : 64;
            int const kInc = mInc ? first_lane(mInc) >> 1 : 64;
            int K = 32;
            if (kEnd + 1 < K) K = kEnd + 1;
            if (kInc + 1 < K) K = kInc + 1;
            unsigned long long const liveMask = below_mask(2 * K);
            bool const live = (int)lane < 2 * K;

            // table gather; the slot then doubles as the duplicate detector: every live lane leaves its lane id in
            // lo[h] and reads it back — a lane that reads another id shares its hash with a lane of this batch.
            // Each live lane rewrites its slot below (new position, or the old value), so nothing leaks.
            uint32_t const cur32 = (uint32_t)cur.bytes;
            uint32_t const h = hash_pos<MLS>(cur.bytes, hshift);
            uint32_t old = T.lo[h];
            if (ip0 > 65536) old |= ((T.hi[h >> 5] >> (h & 31)) & 1u) << 16;
            __builtin_amdgcn_wave_barrier();
            if (live) T.lo[h] = (uint16_t)lane;
            __builtin_amdgcn_wave_barrier();
            uint32_t const back = T.lo[h];
            ZPROF_COUNT(10, 1);

            // speculative loads for the next batch (used if this one has no event)
            uint32_t const nip0 = ip0 + g0 + (uint32_t)(K - 1) * step;
            uint32_t const nstep = step + (uint32_t)(K - 1 == kInc);
            uint32_t nposOff = posOff, nrposOff = rposOff;
            if (g0 != step || nstep != step) batch_offsets(step, nstep, nposOff, nrposOff);
            FastBatch const nxt = batch_load(src, nm8, nip0, nposOff, nrposOff, rep1);

            uint32_t cb = ld32(src + old);                                   // old == 0 reads the unit's first bytes: harmless
            uint32_t cand = old;
            unsigned long long const dupMask = __ballot(back != lane) & liveMask;
            ZPROF(1);
            unsigned long long grp = 0;
            if (dupMask) {
                // exact groups of live lanes with equal hash; a lane's candidate is its closest earlier group member
                unsigned long long ML = dupMask;
                while (ML) {
                    int const j = first_lane(ML);
                    uint32_t const hj = __builtin_amdgcn_readlane(h, j);
                    unsigned long long const G = __ballot(h == hj) & liveMask;
                    if (h == hj) grp = G;
                    ML &= ~G;
                }
                unsigned long long const prevMask = grp & below_mask((int)lane);
                uint32_t const pd = prevMask ? 63u - (uint32_t)__clzll((long long)prevMask) : lane;
                uint32_t const dpos = __shfl(pos, (int)pd), d32 = __shfl(cur32, (int)pd);
                if (prevMask) { cand = dpos; cb = d32; }
            }
            ZPROF(2);
            unsigned long long const mMask = __ballot(cand != 0 && cb == cur32) & liveMask;
            unsigned long long const rMask = rep1 ? (__ballot(cur.rcur == cur.rv) & liveMask & evenLanes) : 0ull;
            int const jm = mMask ? first_lane(mMask) : 64, jr = rMask ? first_lane(rMask) : 64;
            ZPROF(3);
            int const rankM = jm < 64 ? 3 * (jm >> 1) + 1 + (jm & 1) : 0x7fffffff;
            int const rankR = jr < 64 ? 3 * (jr >> 1) : 0x7fffffff;
            int Lcommit;
            if (rankR < rankM)      { evKind = 2; Lcommit = jr + 2; }
            else if (jm < 64)       { evKind = 1; Lcommit = (jm & 1) ? jm + 1 : jm + 2; }
            else                    { evKind = 0; Lcommit = 2 * K; }
            // inserts of the iterations before the event, in lane order (the last lane of a hash group wins); the
            // other live lanes put the old value back
            bool const inC = (int)lane < Lcommit;
            bool we = live;
            if (dupMask) {
                unsigned long long const inside = grp & below_mask(Lcommit);
                unsigned long long const later = inside & ~below_mask((int)lane + 1);
                we = live && (inC ? later == 0 : inside == 0);
            }
            if (we) T.lo[h] = (uint16_t)(inC ? pos : old);
            if (nip0 > 65536) {
                if (we && inC && (pos >> 16)) __hip_atomic_fetch_or(&T.hi[h >> 5], 1u << (h & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_wave_barrier();
            ZPROF(4);

            if (evKind == 1) {
                mpos = __builtin_amdgcn_readlane(pos, jm);
                cand0 = __builtin_amdgcn_readlane(cand, jm);
                cur0 = mpos;
                if ((jm & 1) && step <= 4) {                                 // :318-324 hashTable[hash1] = ip1 (= A_{k+1})
                    // A_{k+1} is lane jm+1's position (its hash is at hand); for jm == 63 it opens the next batch
                    if (jm < 63) { if ((int)lane == jm + 1) tab_put(T, h, pos); }
                    else if (lane == 0) tab_put(T, hash_pos<MLS>(nxt.bytes, hshift), nip0);
                    __builtin_amdgcn_wave_barrier();
                }
                break;
            }
            if (evKind == 2) {
                mpos = __builtin_amdgcn_readlane(rpos, jr);
                cur0 = __builtin_amdgcn_readlane(pos, jr);
                break;
            }
            // no event in this batch: advance like the end of iteration K-1 (:336-348)
            ip0 = nip0; g0 = step;
            if (K - 1 == kEnd) break;                                        // ip3 >= ilimit: unit finished
            if (K - 1 == kInc) { step++; nextStep += 128; }
            posOff = nposOff; rposOff = nrposOff;
            cur = nxt;
        }
        if (evKind == 0) break;

        // ---- _offset / _match (:377-401)
        uint32_t offBase, lim;
        if (evKind == 1) {
            rep2 = rep1; rep1 = mpos - cand0;
            offBase = rep1 + 3;
            lim = (mpos - anchor) < cand0 ? (mpos - anchor) : cand0;
        } else {
            cand0 = mpos - rep1;
            offBase = 1;
            lim = 1;                                                         // :271 mLength = ip0[-1] == match0[-1]
        }
        uint32_t backLen, fwdLen;
        ZPROF(5);
        wave_extend(src, nm8, mpos, cand0, lim, backLen, fwdLen);
        ZPROF(6);
        ZPROF_COUNT(11, 1);
        ip0 = mpos - backLen;
        {   uint32_t const mLength = 4 + backLen + fwdLen;
            lits_copy(out, src, nm8, anchor, ip0 - anchor);
            store_seq(out, ip0 - anchor, offBase, mLength);
            ip0 += mLength; anchor = ip0;
        }

        // ---- :403-420 complementary inserts + immediate repcode; the next batch's bytes ride along.
        // One round of loads: lanes 0..61 compare 8 bytes at ip0+8*lane with the bytes rep2 back, lane 62 fetches
        // the bytes of current0+2, lane 63 those of ip0-2 (the two inserts of :407-408).
        if ((int32_t)ip0 <= ilimit) {
            bool first = true;
            for (;;) {
                uint32_t const q = ip0 + 8u * lane;
                uint32_t qc = q < nm8 ? q : nm8;
                uint32_t const sh = q - qc;
                if (first) { if (lane == 62) qc = cur0 + 2; if (lane == 63) qc = ip0 - 2; }
                uint64_t const a = ld64(src + qc);
                uint64_t x = a ^ ld64(src + (qc - rep2));
                FastBatch const nxt = batch_load(src, nm8, ip0, startPosOff, startRposOff, rep1);
                uint32_t const hh = hash_pos<MLS>(a, hshift);
                if (first) {
                    if (lane == 62) tab_put(T, hh, cur0 + 2);
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 63) tab_put(T, hh, ip0 - 2);
                    __builtin_amdgcn_wave_barrier();
                }
                uint32_t rLength = 0;
                if (rep2 > 0) {
                    x >>= 8 * (sh & 7);
                    uint32_t const s = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8 - sh;
                    uint32_t const same = sh >= 8 ? 0 : s;
                    unsigned long long const stop = __ballot(same < 8) & below_mask(62);
                    if (stop) { int const f = first_lane(stop); rLength = 8u * (uint32_t)f + __builtin_amdgcn_readlane(same, f); }
                    else rLength = 496 + wave_count_fwd(src, ip0 + 496, ip0 + 496 - rep2, nm8);
                }
                if (rLength < 4) { cur = nxt; have = true; break; }          // :411 MEM_read32(ip0) != MEM_read32(ip0 - rep2)
                {   uint32_t const t = rep2; rep2 = rep1; rep1 = t; }
                if (lane == 0) tab_put(T, hh, ip0);                          // lane 0 hashed the bytes at ip0
                __builtin_amdgcn_wave_barrier();
                ip0 += rLength;
                store_seq(out, 0, 1, rLength);
                anchor = ip0;
                first = false;
                if ((int32_t)ip0 > ilimit) break;
            }
        }
        ZPROF(7);
    }
    lits_copy(out, src, nm8, anchor, n - anchor);                           // trailing literals (zstd_compress.c:3365)
    lits_flush(out);
    } else {
        for (uint32_t i = lane; i < n; i += 64) lits[i] = src[i];           // tiny unit: everything is a literal
        out.litPos = n;
    }
    // ---- _cleanup (:368-375)
    ZPROF(8);
    ZPROF_FLUSH(0);
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    if (lane == 0) {
        meta->nbSeq = out.nbSeq; meta->lastLits = n - anchor;
        meta->longPos = out.longPos; meta->longType = out.longType;
        meta->rep[0] = rep1 ? rep1 : saved1; meta->rep[1] = rep2 ? rep2 : saved2; meta->rep[2] = 8;
        meta->status = 0; meta->litSize = out.litPos; meta->pad0 = 0;
    }
}

}  // namespace zhip
