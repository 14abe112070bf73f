// zhip_parse.h — gfx950 match finder for strategy ZSTD_fast, one wavefront per 128 KB unit.
//
// WHAT it computes: exactly the sequences the reference's ZSTD_compressBlock_fast_noDict_generic
// (lib/compress/zstd_fast.c:192-423) emits for a unit with no history (fresh table, rep = {1,4,8}).
//
// HOW (CDNA4 design, not a translation): the reference walks positions one or two at a time because each lookup
// sees the table writes of the positions before it.  Here one 64-lane wavefront owns the unit and evaluates a
// *batch* of up to 62 consecutive search positions at once:
//   * the positions the reference would visit from the current point are a data-independent schedule
//     (pairs A_k, A_k+1 with a step that grows every 128 bytes, zstd_fast.c:232-347); lane j takes the j-th one;
//   * every lane hashes its position, gathers the table entry from LDS (the table lives in LDS: 4 B x 2^hashLog),
//     loads the candidate's 4 bytes from the source in HBM/L2 and tests it; A-lanes also test the repcode;
//   * ballots give the first event in the reference's own order (repcode at ip2, match at ip0, match at ip1);
//   * only the lanes the reference would have inserted before that event write the table.  A later lane must see
//     the insert of an earlier lane with the same hash: that case is detected with a write/read-back on the table
//     itself and resolved by committing the conflict-free prefix and re-gathering (rare);
//   * match extension (backward + forward) is one 512-byte wide compare across the wave.
// All control flow is wave-uniform (derived from ballots), LDS traffic is wave-private, so no s_barrier is needed.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"

namespace zhip {

// ------------------------------------------------------------------ unaligned source access (HBM through L1/L2)
__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
__device__ __forceinline__ uint64_t ld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }   // -> SGPR
__device__ __forceinline__ int first_lane(unsigned long long m) { return __ffsll((long long)m) - 1; }

// multiplicative hashes of lib/compress/zstd_compress_internal.h:820-862, evaluated with 32-bit multiplies:
// only the top hBits of the low 64 bits of the product are needed.
__device__ __forceinline__ uint32_t mulhi64_top32(uint64_t x, uint64_t p)
{   // high 32 bits of the low 64 bits of x*p
    uint32_t const xl = (uint32_t)x, xh = (uint32_t)(x >> 32), pl = (uint32_t)p, ph = (uint32_t)(p >> 32);
    return __umulhi(xl, pl) + xh * pl + xl * ph;
}
__device__ __forceinline__ uint32_t hash_pos(uint64_t bytes, uint32_t hBits, uint32_t mls)
{
    switch (mls) {
    default:
    case 4: return ((uint32_t)bytes * 2654435761U) >> (32 - hBits);
    case 5: return mulhi64_top32(bytes << 24, 889523592379ULL) >> (32 - hBits);
    case 6: return mulhi64_top32(bytes << 16, 227718039650203ULL) >> (32 - hBits);
    case 7: return mulhi64_top32(bytes << 8, 58295818150454627ULL) >> (32 - hBits);
    case 8: return mulhi64_top32(bytes, 0xCF1BBCDCB7A56463ULL) >> (32 - hBits);
    }
}

// Wave-private LDS cells that one lane writes and another lane of the same wave reads back: DS instructions of a
// wave execute in order, so only the compiler must be kept from forwarding/reordering -> volatile accesses.
__device__ __forceinline__ uint32_t lds_get(const uint32_t* T, uint32_t i) { return ((const volatile uint32_t*)T)[i]; }
__device__ __forceinline__ void lds_put(uint32_t* T, uint32_t i, uint32_t v) { ((volatile uint32_t*)T)[i] = v; }

// ------------------------------------------------------------------ wave-wide match extension
// Common-prefix length of src[a..) and src[b..) (b < a), a bounded by n — ZSTD_count (zstd_compress_internal.h:771).
// 64 lanes x 8 bytes per round.
__device__ __forceinline__ uint32_t wave_count_fwd(const uint8_t* src, uint32_t a, uint32_t b, uint32_t n)
{
    int const lane = lane_id();
    uint32_t total = 0;
    for (;;) {
        uint32_t const q = a + 8u * (uint32_t)lane;
        uint32_t const room = q < n ? n - q : 0;           // valid bytes at q
        uint32_t same;                                     // equal leading bytes of this lane's 8-byte window
        if (room >= 8) {
            uint64_t const x = ld64(src + q) ^ ld64(src + (q - (a - b)));
            same = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
        } else {
            same = 0;
            while (same < room && src[q + same] == src[q - (a - b) + same]) same++;
        }
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
    int const lane = lane_id();
    uint32_t total = 0;
    for (;;) {
        uint32_t const i = total + (uint32_t)lane;
        bool const stopHere = (i >= limit) || (src[ip - 1 - i] != src[m - 1 - i]);
        unsigned long long const stop = __ballot(stopHere);
        if (stop) return total + (uint32_t)first_lane(stop);
        total += 64;
    }
}

// ------------------------------------------------------------------ the parser
struct FastOut {
    ZhipSeq* seqs;          // global, capacity ZHIP_SEQ_CAP
    uint32_t nbSeq, longPos, longType;
};

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

// T: wave-private hash table in LDS, 1<<hlog entries, value = position (0 = empty; position 0 is never inserted).
__device__ inline void parse_fast_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u,
                                       uint32_t* T, ZhipSeq* seqs, ZhipParse* meta)
{
    int const lane = lane_id();
    uint32_t const hlog = u.hashLog, mls = u.minMatch;
    uint32_t const stepSize = u.targetLength + !u.targetLength + 1;         // zstd_fast.c:200
    FastOut out; out.seqs = seqs; out.nbSeq = 0; out.longPos = 0; out.longType = 0;

    for (uint32_t i = (uint32_t)lane; i < (1u << hlog); i += 64) lds_put(T, i, 0);     // fresh table (zstd_compress.c:2020)
    __builtin_amdgcn_wave_barrier();

    uint32_t anchor = 0, rep1 = 1, rep2 = 4, saved1 = 0, saved2 = 0;
    {
        int32_t const ilimit = (int32_t)n - 8;                               // may be negative for tiny units
        uint32_t ip0 = 1;
        // :238-244  ip0 = 1, lowest index 0 -> maxRep = 1
        if (rep2 > 1) { saved2 = rep2; rep2 = 0; }
        if (rep1 > 1) { saved1 = rep1; rep1 = 0; }

        bool more = true;
        while (more) {                                                       // one turn per `_start`
            uint32_t step = stepSize, g0 = stepSize, nextStep = ip0 + 128;
            if ((int32_t)(ip0 + g0 + 1) >= ilimit) break;                    // :257
            // ---- scan batches until an event or the end of the unit
            int evKind = 0;                      // 0 none (unit finished), 1 match, 2 repcode
            uint32_t evPos = 0, evCand = 0, cur0 = 0;
            for (;;) {
                // how many reference iterations (pairs) this batch covers
                int32_t const X = ilimit - 1 - (int32_t)(ip0 + g0);          // end:  (k+1)*step >= X
                int32_t const Y = (int32_t)nextStep - (int32_t)(ip0 + g0);   // step++: (k+1)*step >= Y
                int32_t kEnd = X <= 0 ? 0 : (int32_t)((X + (int32_t)step - 1) / (int32_t)step) - 1;
                int32_t kInc = Y <= 0 ? 0 : (int32_t)((Y + (int32_t)step - 1) / (int32_t)step) - 1;
                int32_t K = 31;
                if (kEnd + 1 < K) K = kEnd + 1;
                if (kInc + 1 < K) K = kInc + 1;
                int const nLanes = 2 * K;                                    // search lanes; lane 2K = repcode-only

                int const k = lane >> 1;
                uint32_t const pos = (k == 0 ? ip0 : ip0 + g0 + (uint32_t)(k - 1) * step) + (uint32_t)(lane & 1);
                bool const live = lane <= nLanes;
                uint64_t const bytes = live ? ld64(src + pos) : 0;
                uint32_t const cur32 = (uint32_t)bytes;
                uint32_t const h = hash_pos(bytes, hlog, mls);
                bool const isRepLane = live && ((lane & 1) == 0) && lane >= 2 && rep1 > 0;
                uint32_t const rv = isRepLane ? ld32(src + pos - rep1) : 0;
                unsigned long long const repMask = __ballot(isRepLane && rv == cur32);
                int const jr = repMask ? first_lane(repMask) : 64;
                int const rankR = jr < 64 ? 3 * ((jr >> 1) - 1) : 0x7fffffff;

                int done = 0, Ltest = 0, Lcommit = 0, jm = 64;
                uint32_t old = 0;
                for (;;) {                                                   // conflict-resolution passes (usually 1)
                    bool const act = live && lane >= done;
                    old = act ? lds_get(T, h) : 0;
                    bool hit = false;
                    if (act && lane < nLanes && old != 0) hit = (ld32(src + old) == cur32);
                    unsigned long long const mMask = __ballot(hit);
                    jm = mMask ? first_lane(mMask) : 64;
                    int const rankM = jm < 64 ? 3 * (jm >> 1) + 1 + (jm & 1) : 0x7fffffff;
                    if (rankR < rankM)      { evKind = 2; Ltest = jr - 2; Lcommit = jr; }
                    else if (jm < 64)       { evKind = 1; Ltest = jm + 1; Lcommit = (jm & 1) ? jm + 1 + (step <= 4 ? 1 : 0) : jm + 2; }
                    else                    { evKind = 0; Ltest = nLanes; Lcommit = nLanes; }
                    bool const inC = lane >= done && lane < Lcommit;
                    if (inC) lds_put(T, h, pos);
                    __builtin_amdgcn_wave_barrier();
                    uint32_t const back = inC ? lds_get(T, h) : pos;
                    if (!__ballot(inC && back != pos)) break;                // no two committed lanes share a slot
                    // ---- rare: two lanes of the committed range hash alike.  Undo, find the first duplicate.
                    if (inC) lds_put(T, h, old);
                    __builtin_amdgcn_wave_barrier();
                    int jstar = Lcommit;
                    for (int x = done + 1; x < Lcommit; x++) {
                        uint32_t const hx = __builtin_amdgcn_readlane(h, x);
                        if (__ballot(lane >= done && lane < x && h == hx)) { jstar = x; break; }
                    }
                    if (lane >= done && lane < jstar) lds_put(T, h, pos);            // conflict-free prefix
                    __builtin_amdgcn_wave_barrier();
                    if (jstar >= Ltest) {                                    // only insert-only lanes collide: keep order
                        for (int x = jstar; x < Lcommit; x++) {
                            if (lane == x) lds_put(T, h, pos);
                            __builtin_amdgcn_wave_barrier();
                        }
                        break;
                    }
                    done = jstar;                                            // lane jstar now sees its true candidate
                }
                if (evKind == 1) {
                    evPos = __builtin_amdgcn_readlane(pos, jm);
                    evCand = __builtin_amdgcn_readlane(old, jm);
                    cur0 = evPos;
                    break;
                }
                if (evKind == 2) {
                    evPos = __builtin_amdgcn_readlane(pos, jr);
                    cur0 = __builtin_amdgcn_readlane(pos, jr - 2);
                    break;
                }
                // no event in this batch: advance like the end of iteration K-1 (:336-348)
                ip0 = ip0 + g0 + (uint32_t)(K - 1) * step;
                g0 = step;
                if (K - 1 == kEnd) break;                                    // ip3 >= ilimit: unit finished
                if (K - 1 == kInc) { step++; nextStep += 128; }
            }
            if (evKind == 0) break;

            // ---- _offset / _match (:377-401)
            uint32_t mLength, offBase, match0;
            ip0 = evPos;
            if (evKind == 1) {
                match0 = evCand;
                rep2 = rep1; rep1 = ip0 - match0;
                offBase = rep1 + 3;
                uint32_t const lim = (ip0 - anchor) < match0 ? (ip0 - anchor) : match0;
                uint32_t const backLen = wave_count_back(src, ip0, match0, lim);
                ip0 -= backLen; match0 -= backLen;
                mLength = 4 + backLen;
            } else {
                match0 = ip0 - rep1;
                uint32_t const b1 = uni((uint32_t)(src[ip0 - 1] == src[match0 - 1]));
                ip0 -= b1; match0 -= b1;
                offBase = 1;
                mLength = 4 + b1;
            }
            mLength += wave_count_fwd(src, ip0 + mLength, match0 + mLength, n);
            store_seq(out, ip0 - anchor, offBase, mLength);
            ip0 += mLength; anchor = ip0;

            // ---- :403-420 complementary inserts + immediate repcode
            if ((int32_t)ip0 <= ilimit) {
                uint32_t const pA = cur0 + 2, pB = ip0 - 2;
                uint64_t const by = ld64(src + (lane == 0 ? pA : pB));
                uint32_t const hh = hash_pos(by, hlog, mls);
                if (lane == 0) lds_put(T, hh, pA);
                __builtin_amdgcn_wave_barrier();
                if (lane == 1) lds_put(T, hh, pB);
                __builtin_amdgcn_wave_barrier();
                if (rep2 > 0) {
                    while ((int32_t)ip0 <= ilimit) {
                        uint64_t const b0 = ld64(src + ip0);
                        if ((uint32_t)b0 != ld32(src + ip0 - rep2)) break;
                        uint32_t const rLength = wave_count_fwd(src, ip0 + 4, ip0 + 4 - rep2, n) + 4;
                        uint32_t const t = rep2; rep2 = rep1; rep1 = t;
                        if (lane == 0) lds_put(T, hash_pos(b0, hlog, mls), ip0);
                        __builtin_amdgcn_wave_barrier();
                        ip0 += rLength;
                        store_seq(out, 0, 1, rLength);
                        anchor = ip0;
                    }
                }
            }
        }
    }
    // ---- _cleanup (:368-375)
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    if (lane == 0) {
        meta->nbSeq = out.nbSeq; meta->lastLits = n - anchor;
        meta->longPos = out.longPos; meta->longType = out.longType;
        meta->rep[0] = rep1 ? rep1 : saved1; meta->rep[1] = rep2 ? rep2 : saved2; meta->rep[2] = 8;
        meta->status = 0;
    }
}

}  // namespace zhip
