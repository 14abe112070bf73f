// zhip_parse.h — gfx950 match finder for strategy ZSTD_fast, one wavefront per 128 KB unit (or per block of a frame).
//
// WHAT it computes: exactly the sequences the reference's ZSTD_compressBlock_fast_noDict_generic
// (lib/compress/zstd_fast.c:192-423) emits for a unit with no history (fresh table, rep = {1,4,8}).
//
// HOW (CDNA4 design, not a translation).  The reference walks positions one or two at a time because each lookup
// sees the table writes of the positions before it.  Here one 64-lane wavefront owns the unit; the table is wave-private LDS
// (16-bit entries + a 1-bit plane for bit 16 of the position = 17 KB for hashLog 13, so NINE units are resident per CU: LDS
// is allocated in 1280-byte granules on gfx950, 9 x 14 granules).  Two scan shapes share it:
//   * the dense-scan WINDOW (window_batch, below): while the gap between searched pairs is 2 every position is searched, so
//     lane l takes position B+l and ONE table gather + ONE candidate gather resolve every event among 60 positions — matches,
//     repcodes, extensions, the immediate-repcode loop, inserts and literal stores — with mask arithmetic.  This is where
//     dense-match data spends its time;
//   * the schedule-shaped batch (parse_fast_block's second path, the round-1 design): for the stretches where the gap has
//     grown (long literal runs), lane 2k+b takes the k-th pair of the reference's data-independent schedule (pairs A_k, A_k+1
//     with a gap that grows every 128 bytes, zstd_fast.c:232-347), ballots give the first event in the reference's own order
//     (repcode at ip2, match at ip0, match at ip1), extension in one wave-wide compare (48 x 8 B forward, 16 x 8 B backward),
//     then one round that fetches the bytes of the two complementary inserts, the immediate-repcode probe + count and the next
//     batch;
//   * in both, lanes that hash alike must see each other's inserts in lane order: the slot itself is the detector (every lane
//     leaves its lane id in its slot and reads it back), ballots turn the flagged lanes into exact per-hash groups.
// The parser is written against a table policy (tab_get / tab_put / tab_mark / tab_peek / tab_unmark) and takes a block range,
// the lowest valid match position and the incoming repcodes, so the same code parses the blocks of a multi-block frame on a
// table of 32-bit positions (zhip_frame.h).
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
__device__ __forceinline__ unsigned long long uni64(unsigned long long v)
{
    return (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)v) | ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
}
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
// entry of slot h; `high`: positions >= 65536 may exist (only then is the bit plane read)
__device__ __forceinline__ uint32_t tab_get(const FastTab& T, uint32_t h, bool high)
{
    uint32_t v = T.lo[h];
    if (high) v |= ((T.hi[h >> 5] >> (h & 31)) & 1u) << 16;
    return v;
}
// the slot as duplicate detector: leave a small marker (a lane id), read it back, put the old entry back
__device__ __forceinline__ void tab_mark(const FastTab& T, uint32_t h, uint32_t v) { T.lo[h] = (uint16_t)v; }
__device__ __forceinline__ uint32_t tab_peek(const FastTab& T, uint32_t h) { return T.lo[h]; }
__device__ __forceinline__ void tab_unmark(const FastTab& T, uint32_t h, uint32_t old) { T.lo[h] = (uint16_t)old; }

// ---- TAGS (round 4).  A window fetches the 4 bytes at every lane's table candidate — 64 random lines for 64 positions of source, of which
// about one is a match: the stage's HBM traffic was 10-20x its algorithmic bytes and each such gather costs four coalesced loads' time.
// A tag is a second hash of exactly the 4 bytes the parser compares at a candidate (MEM_read32(match) == MEM_read32(ip), zstd_fast.c:289):
// different tags PROVE the compare fails, so the candidate's bytes are only fetched by the lanes whose tag agrees.  The unit table in LDS
// has room for 2 bits per entry (a third bit plane pair; 8 units per CU instead of 9), the table in global memory carries 15 bits above
// the 17-bit position.  Every insert writes the tag of the bytes at the inserted position; tables without room (frames) report "maybe".
#ifndef ZHIP_FAST_TAGS
#define ZHIP_FAST_TAGS 1             /* 0: no tags, the tables as they were (A/B builds) */
#endif
__device__ __forceinline__ uint32_t fast_tag15(uint32_t b4) { return (b4 * 0x9E3779B1u) >> 17; }
struct FastTagTab { lds_u16* lo; lds_u32* hi; lds_u32* tg; };
__host__ __device__ inline uint32_t fast_tg_bytes(uint32_t hlog) { uint32_t const b = (1u << hlog) >> 2; return ZHIP_FAST_TAGS ? (b < 4 ? 4 : b) : 0u; }
__host__ __device__ inline uint32_t fast_tag_lds_bytes(uint32_t hlog) { return fast_lds_bytes(hlog) + fast_tg_bytes(hlog); }
__device__ __forceinline__ void tab_put_t(const FastTagTab& T, uint32_t h, uint32_t pos, uint32_t tag)
{
    T.lo[h] = (uint16_t)pos;
    if (pos >> 16) __hip_atomic_fetch_or(&T.hi[h >> 5], 1u << (h & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (ZHIP_FAST_TAGS) {
        uint32_t const sh = 2u * (h & 15u);
        __hip_atomic_fetch_and(&T.tg[h >> 4], ~(3u << sh), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_or(&T.tg[h >> 4], (tag >> 13) << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}
__device__ __forceinline__ uint32_t tab_get_t(const FastTagTab& T, uint32_t h, bool high, uint32_t tag, bool& maybe)
{
    uint32_t v = T.lo[h];
    if (high) v |= ((T.hi[h >> 5] >> (h & 31)) & 1u) << 16;
    maybe = !ZHIP_FAST_TAGS || ((T.tg[h >> 4] >> (2u * (h & 15u))) & 3u) == (tag >> 13);
    return v;
}
__device__ __forceinline__ void tab_mark(const FastTagTab& T, uint32_t h, uint32_t v) { T.lo[h] = (uint16_t)v; }
__device__ __forceinline__ uint32_t tab_peek(const FastTagTab& T, uint32_t h) { return T.lo[h]; }
__device__ __forceinline__ void tab_unmark(const FastTagTab& T, uint32_t h, uint32_t old) { T.lo[h] = (uint16_t)old; }

// the same table with full 32-bit positions, for blocks that share one table across a multi-block frame (positions are relative
// to the frame start and exceed 2^17); a flat pointer: LDS when 4 << hashLog bytes fit, else global memory
struct WideTab { uint32_t* w; };
__device__ __forceinline__ void tab_put(const WideTab& T, uint32_t h, uint32_t pos) { T.w[h] = pos; }
__device__ __forceinline__ uint32_t tab_get(const WideTab& T, uint32_t h, bool) { return T.w[h]; }
__device__ __forceinline__ void tab_mark(const WideTab& T, uint32_t h, uint32_t v) { T.w[h] = v; }
__device__ __forceinline__ uint32_t tab_peek(const WideTab& T, uint32_t h) { return T.w[h]; }
__device__ __forceinline__ void tab_unmark(const WideTab& T, uint32_t h, uint32_t old) { T.w[h] = old; }
__device__ __forceinline__ void tab_put_t(const WideTab& T, uint32_t h, uint32_t pos, uint32_t) { T.w[h] = pos; }
__device__ __forceinline__ uint32_t tab_get_t(const WideTab& T, uint32_t h, bool, uint32_t, bool& maybe) { maybe = true; return T.w[h]; }

// 24-bit positions in LDS for the blocks of a frame or job whose positions stay below 2^24: 16 low bits + one byte, 3 bytes per
// entry instead of WideTab's 4 (48 KB at hashLog 14: two frame workgroups per CU instead of one) and LDS-typed pointers (ds_ instead of
// flat_ instructions).  The marker of the duplicate detector only ever touches lo[], like FastTab's.
struct Lds24Tab { lds_u16* lo; lds_u8* hi; };
__device__ __forceinline__ void tab_put(const Lds24Tab& T, uint32_t h, uint32_t pos) { T.lo[h] = (uint16_t)pos; T.hi[h] = (uint8_t)(pos >> 16); }
__device__ __forceinline__ uint32_t tab_get(const Lds24Tab& T, uint32_t h, bool) { return (uint32_t)T.lo[h] | ((uint32_t)T.hi[h] << 16); }
__device__ __forceinline__ void tab_mark(const Lds24Tab& T, uint32_t h, uint32_t v) { T.lo[h] = (uint16_t)v; }
__device__ __forceinline__ uint32_t tab_peek(const Lds24Tab& T, uint32_t h) { return T.lo[h]; }
__device__ __forceinline__ void tab_unmark(const Lds24Tab& T, uint32_t h, uint32_t old) { T.lo[h] = (uint16_t)old; }
__device__ __forceinline__ void tab_put_t(const Lds24Tab& T, uint32_t h, uint32_t pos, uint32_t) { tab_put(T, h, pos); }
__device__ __forceinline__ uint32_t tab_get_t(const Lds24Tab& T, uint32_t h, bool high, uint32_t, bool& maybe) { maybe = true; return tab_get(T, h, high); }

// the unit table with 32-bit positions in GLOBAL memory (L2 / Infinity Cache), for the wavefronts that run beside the LDS-table ones on a
// CU whose LDS is full (k_parse_fast_g): no LDS at all, so the slot cannot double as the duplicate detector (three dependent global
// round trips) — lanes that hash alike are found with one ballot per hash bit instead (wave_hash_group; TabTraits<>::ballotGroups)
struct GlobTab { uint32_t* w; };
__device__ __forceinline__ void tab_put(const GlobTab& T, uint32_t h, uint32_t pos) { T.w[h] = pos; }
__device__ __forceinline__ uint32_t tab_get(const GlobTab& T, uint32_t h, bool) { return T.w[h]; }
__device__ __forceinline__ void tab_mark(const GlobTab&, uint32_t, uint32_t) { }
__device__ __forceinline__ uint32_t tab_peek(const GlobTab&, uint32_t) { return 0; }
__device__ __forceinline__ void tab_unmark(const GlobTab&, uint32_t, uint32_t) { }
__device__ __forceinline__ void tab_put_t(const GlobTab& T, uint32_t h, uint32_t pos, uint32_t tag) { T.w[h] = ZHIP_FAST_TAGS ? (pos | (tag << 17)) : pos; }      // positions are < 2^17 here (units)
__device__ __forceinline__ uint32_t tab_get_t(const GlobTab& T, uint32_t h, bool, uint32_t tag, bool& maybe)
{
    uint32_t const e = T.w[h];
    maybe = !ZHIP_FAST_TAGS || (e >> 17) == tag;
    return ZHIP_FAST_TAGS ? (e & 0x1FFFFu) : e;
}
template <typename TAB> struct TabTraits { static constexpr bool ballotGroups = false; };
template <> struct TabTraits<GlobTab> { static constexpr bool ballotGroups = true; };
// per lane: the lanes of the wavefront whose `bits`-bit value equals this lane's (itself included) — one ballot per bit
__device__ __forceinline__ unsigned long long wave_hash_group(uint32_t h, uint32_t bits)
{
    uint32_t glo = ~0u, ghi = ~0u;
    for (uint32_t b = 0; b < bits; b++) {
        bool const bit = (h >> b) & 1u;
        unsigned long long const S = __ballot(bit);
        uint32_t const flip = bit ? 0u : ~0u;
        glo &= (uint32_t)S ^ flip; ghi &= (uint32_t)(S >> 32) ^ flip;
    }
    return (unsigned long long)glo | ((unsigned long long)ghi << 32);
}

// where the speculative read of a candidate's bytes goes when the entry is 0 (= empty; the loaded value is never used then).  The
// unit tables read position 0 — the unit's first bytes, always there.  A job of a frame whose window starts at the frame's byte 0 counts
// its positions from 1 with `src` one byte BEFORE the frame (zhip_frame.h): position 0 is not memory, so the frame tables (WideTab,
// Lds24Tab) read position 1.
__device__ __forceinline__ uint32_t tab_guard(const FastTab&, uint32_t old) { return old; }
__device__ __forceinline__ uint32_t tab_guard(const FastTagTab&, uint32_t old) { return old; }
__device__ __forceinline__ uint32_t tab_guard(const Lds24Tab&, uint32_t old) { return old > 1u ? old : 1u; }
__device__ __forceinline__ uint32_t tab_guard(const WideTab&, uint32_t old) { return old > 1u ? old : 1u; }
__device__ __forceinline__ uint32_t tab_guard(const GlobTab&, uint32_t old) { return old; }

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
__device__ __attribute__((noinline)) uint32_t wave_count_fwd_far(const uint8_t* src, uint32_t a, uint32_t b, uint32_t nm8)
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
__device__ __attribute__((noinline)) uint32_t wave_count_back_far(const uint8_t* src, uint32_t ip, uint32_t m, uint32_t limit)
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

// Both are out of line on purpose.  They are the rare continuation of an extension that left the 64 positions at hand; inlined
// into the parser's event loop they cost it ~45 spilled SGPRs.  A called function returns in a VGPR, and the result is left
// there: what derives from it (lengths, scan position, anchor) then lives in vector registers, which is what relieves the
// scalar register file (measured on MI355X: text 66.8 -> 61.0 ms, Silesia-shaped 39.8 -> 34.7 ms per GiB-scale launch;
// forcing the result back to an SGPR with readfirstlane brings the spills and the old times back).
__device__ __forceinline__ uint32_t wave_count_fwd(const uint8_t* src, uint32_t a, uint32_t b, uint32_t nm8) { return wave_count_fwd_far(src, a, b, nm8); }
__device__ __forceinline__ uint32_t wave_count_back(const uint8_t* src, uint32_t ip, uint32_t m, uint32_t limit) { return wave_count_back_far(src, ip, m, limit); }

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
#ifdef ZHIP_PROF
    uint64_t* zp; uint64_t* zlast;   // the caller's phase accumulators (measurement build only)
#endif
};
#ifdef ZHIP_PROF
#define ZWPROF(o, i) do { uint64_t const t_ = __builtin_amdgcn_s_memtime(); (o).zp[i] += t_ - *(o).zlast; *(o).zlast = t_; } while (0)
#define ZWPROF_COUNT(o, i, v) do { (o).zp[i] += (uint64_t)(v); } while (0)
#define ZWPROF_SYNC(o, i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ZWPROF(o, i); } while (0)   /* charges the outstanding loads to phase i */
#else
#define ZWPROF_SYNC(o, i) do { } while (0)
#define ZWPROF(o, i) do { } while (0)
#define ZWPROF_COUNT(o, i, v) do { } while (0)
#endif

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

// ------------------------------------------------------------------ after a match: zstd_fast.c:403-420 by loads
// The two complementary inserts (when `first`) and the immediate-repcode loop; one round of loads per repcode match:
// lanes 0..61 compare 8 bytes at ip0+8*lane with the bytes rep2 back, lane 62 fetches the bytes of cur0+2, lane 63 those
// of ip0-2 (the two inserts of :407-408).  With NEXT the bytes of the batch that starts at the final ip0 ride along
// (returned in `cur`, result true).  Requires ip0 <= ilimit (= nm8).
template <uint32_t MLS, bool NEXT, typename TAB>
__device__ __forceinline__ bool post_match(const uint8_t* __restrict__ src, uint32_t nm8, uint32_t hshift, const TAB& T, FastOut& out,
                                           uint32_t& ip0, uint32_t& anchor, uint32_t& rep1, uint32_t& rep2, uint32_t cur0, bool first,
                                           uint32_t startPosOff, uint32_t startRposOff, FastBatch& cur)
{
    uint32_t const lane = (uint32_t)lane_id();
    int32_t const ilimit = (int32_t)nm8;
    for (;;) {
        uint32_t const q = ip0 + 8u * lane;
        uint32_t qc = q < nm8 ? q : nm8;
        uint32_t const sh = q - qc;
        if (first) { if (lane == 62) qc = cur0 + 2; if (lane == 63) qc = ip0 - 2; }
        uint64_t const a = ld64(src + qc);
        uint64_t x = a ^ ld64(src + (qc - rep2));
        FastBatch nxt; nxt.bytes = 0; nxt.rcur = 0; nxt.rv = 0;
        if (NEXT) nxt = batch_load(src, nm8, ip0, startPosOff, startRposOff, rep1);
        uint32_t const hh = hash_pos<MLS>(a, hshift);
        if (first) {
            if (lane == 62) tab_put_t(T, hh, cur0 + 2, fast_tag15((uint32_t)a));
            __builtin_amdgcn_wave_barrier();
            if (lane == 63) tab_put_t(T, hh, ip0 - 2, fast_tag15((uint32_t)a));
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
        if (rLength < 4) { cur = nxt; return NEXT; }                     // :411 MEM_read32(ip0) != MEM_read32(ip0 - rep2)
        {   uint32_t const t = rep2; rep2 = rep1; rep1 = t; }
        if (lane == 0) tab_put_t(T, hh, ip0, fast_tag15((uint32_t)a));   // lane 0 hashed the bytes at ip0
        __builtin_amdgcn_wave_barrier();
        ip0 += rLength;
        store_seq(out, 0, 1, rLength);
        anchor = ip0;
        first = false;
        if ((int32_t)ip0 > ilimit) return false;
    }
}

// post_match for the window parser's rare case (a match that runs past the window), out of line for the same reason as the
// wave_count_*_far pair: what it needs goes in and comes back by value
struct PostState { uint32_t ip0, anchor, rep1, rep2, nbSeq, longPos, longType; };
template <uint32_t MLS, typename TAB>
__device__ __attribute__((noinline)) PostState post_match_far(const uint8_t* src, uint32_t nm8, uint32_t hshift, TAB T, ZhipSeq* seqs,
                                                              PostState st, uint32_t cur0, bool first)
{
    FastOut o; o.seqs = seqs; o.lits = nullptr; o.nbSeq = st.nbSeq; o.longPos = st.longPos; o.longType = st.longType;
    o.litPos = 0; o.pendV = 0; o.pendSh = 0; o.pendOff = 0; o.pendLen = 0;
    FastBatch dummy;
    post_match<MLS, false, TAB>(src, nm8, hshift, T, o, st.ip0, st.anchor, st.rep1, st.rep2, cur0, first, 0, 0, dummy);
    st.nbSeq = o.nbSeq; st.longPos = o.longPos; st.longType = o.longType;
    return st;
}

// ------------------------------------------------------------------ the dense-scan WINDOW: many events per gather
// While the gap between searched pairs is 2 (the reference restarts at 2 after every match and keeps it for 128 bytes,
// zstd_fast.c:232-236, :342-347) every position is searched, so lane l simply takes position B+l.  One window costs one
// table gather (LDS) and one candidate gather (HBM/L2), and then resolves EVERY event among its lanes in the reference's
// order with wave-uniform mask arithmetic:
//   * M   — lanes whose table candidate matches 4 bytes (valid for the lanes whose hash no earlier lane of the window
//           shares: their candidate cannot depend on what the window itself inserts; a lane that does share one is resolved
//           exactly from its two closest earlier group members and the insert mask, the window ends before the first lane with
//           three of them);
//   * E1/E2 — per repcode offset, which lanes equal the byte (…b) / the 4 bytes (…q) that offset back: ONE coalesced
//           load per offset.  Repcode probes (:268), the one-byte backward step of a repcode match (:271), forward and
//           backward extension (:387-391, ZSTD_count) and the immediate-repcode loop (:410-420) are bit scans of these
//           masks; only a match with a NEW offset needs a load (its E mask), and only a match that runs past the window
//           goes back to wave-wide compares;
//   * inserts are collected in a mask and written once (their hashes are distinct by construction); inserts of lanes
//     beyond the exact prefix follow one by one in position order;
//   * literals: every lane not covered by a match stores its own byte at (position - bytes matched so far).
enum { ZW_CONT = 0, ZW_INC = 1, ZW_RESTART = 2 };
#define ZHIP_WIN_NEED 80u            /* a window at B needs B + 80 <= n: 64 positions x 8-byte reads, all iterations inside ilimit */
#ifndef ZHIP_WIN_DENSE_ONLY
#define ZHIP_WIN_DENSE_ONLY 0       /* 1: a scan leaves window mode after its first event-less window */
#endif
#ifndef ZHIP_WIN_GROUPS
#define ZHIP_WIN_GROUPS 16           /* hash groups of a window resolved exactly; the exact prefix ends at the first lane of the next one */
#endif
#define ZHIP_WIN_LANES 60            /* events are taken from lanes below this (their +2/+4 neighbours stay inside the window) */

#ifndef ZHIP_SBFM64                  /* s_bfm_b64: `width` (0..63) lanes from lane `offset` (0..63) on (the emulator brings its own) */
__device__ __forceinline__ unsigned long long zhip_sbfm64(uint32_t width, uint32_t offset)
{
    unsigned long long r;
    asm("s_bfm_b64 %0, %1, %2" : "=s"(r) : "s"(width), "s"(offset));
    return r;
}
#define ZHIP_SBFM64(width, offset) zhip_sbfm64(__builtin_amdgcn_readfirstlane(width), __builtin_amdgcn_readfirstlane(offset))
#endif
#ifndef ZHIP_WRITELANE               /* v_writelane_b32: lane `l` of `old` := the wave-uniform `v` (the emulator brings its own) */
extern "C" __device__ unsigned zhip_llvm_writelane(unsigned, unsigned, unsigned) __asm("llvm.amdgcn.writelane.i32");
#define ZHIP_WRITELANE(v, l, old) zhip_llvm_writelane(v, l, old)
#endif

// lane masks from shifts; every index is in 0..63 by construction (no range checks: this is the serial part of the parser)
__device__ __forceinline__ unsigned long long lanes_from(uint32_t l) { return ~0ull << l; }           // lanes l .. 63
__device__ __forceinline__ unsigned long long lanes_below(uint32_t l) { return ~(~0ull << l); }       // lanes 0 .. l-1
__device__ __forceinline__ uint32_t ff1u(unsigned long long m) { return (uint32_t)(__ffsll((long long)m) - 1); }   // 0xFFFFFFFF when empty

// equal bytes from lane s0 on according to Eb (bit l: src[B+l] == src[B+l-off]); beyond the window by wave-wide compares
__device__ __forceinline__ uint32_t fwd_run(const uint8_t* src, uint32_t nm8, uint32_t B, unsigned long long Eb, uint32_t s0, uint32_t off)
{
    uint32_t base = 0, from = B + s0;
    if (s0 < 64) {
        unsigned long long const inv = ~Eb >> s0;
        if (inv) return ff1u(inv);
        base = 64 - s0; from = B + 64;
    }
    return base + wave_count_fwd(src, from, from - off, nm8);
}

// 16 bytes at an arbitrary address (one global_load_dwordx4: gfx950 runs with unaligned access enabled)
struct Quad { uint32_t x, y, z, w; };
__device__ __forceinline__ Quad ld128(const uint8_t* p) { Quad v; __builtin_memcpy(&v, p, 16); return v; }
// equal bytes of a 4-byte word pair from its low end (0..4), given their XOR
__device__ __forceinline__ uint32_t same_lo(uint32_t x) { uint32_t const t = (uint32_t)(__ffs((int)x) - 1) >> 3; return t < 4u ? t : 4u; }
// highest / lowest set bit of a 64-bit mask held as two words (the mask must not be empty)
__device__ __forceinline__ uint32_t top_bit(uint32_t lo, uint32_t hi) { return hi ? 63u - (uint32_t)__clz((int)hi) : 31u - (uint32_t)__clz((int)lo); }
__device__ __forceinline__ uint32_t low_bit(uint32_t lo, uint32_t hi) { return lo ? (uint32_t)__ffs((int)lo) - 1u : 31u + (uint32_t)__ffs((int)hi); }
__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t pull(uint32_t v, uint32_t srcLane) { return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(srcLane << 2), (int)v); }

// EVENT CHAINS (round 3).  A window's cost used to be per EVENT and serial: a dependent load for the E masks of every new offset
// (they carry the extension of the match AND the repcode probes that follow it) and ~250 wave-uniform instructions — which this
// machine runs at 5-10 cycles each (a wavefront issues one instruction per ~4 cycles, a value that goes from a vector compare into a
// scalar instruction waits ~20: scripts/ubench/issue.hip).  Dense-match data has five events per window.  Now:
//   * every lane fetches 24 bytes around its own table candidate with the candidate gather itself (the same sectors) and resolves its
//     own would-be match in the lane: backward up to 4 bytes, forward up to 16 (`info`);
//   * the first event of a window is still found the exact way (its repcode probes use the masks the front loaded).  If it is a plain
//     match a short scalar CHASE follows the chain of events it starts — next candidate lane at or after the end of the previous
//     match, one v_readlane per hop — ASSUMING that no repcode check in between hits;
//   * everything else about the events of the chain — match start, lengths, sequences, covered lanes, inserted lanes — is computed
//     for ALL of them at once, each event in its own lane (four ds_bpermute in two rounds, no loop);
//   * the assumption is checked afterwards: the lanes where the reference would have probed a repcode (zstd_fast.c:268) or checked
//     the immediate repcode (:410), each with the offset valid at that point, are compared in ONE masked gather.  A hit is rare
//     (repcodes are 1-3 % of the sequences); the window is then run again from its entry state the exact way — nothing of it has left
//     the registers by then except sequence records that the exact pass overwrites;
//   * a match that runs past lane 63 but whose end its lane knows needs no loads either: the next window starts two positions in
//     front of that end (CARRY), so that the complementary insert of end-2 (:408) and the immediate-repcode check at the end are lanes
//     0 and 2 of an ordinary window.
#ifndef ZHIP_WIN_FAST
#define ZHIP_WIN_FAST 0              /* 0: every event the exact way.  1: event chains — byte-identical, measured slower so far */
#endif
#ifndef ZHIP_FAST_TAIL
#define ZHIP_FAST_TAIL 16u           /* a chain of one event: an unverified tail shorter than this is left to the next window instead of being verified by a load */
#endif
#define ZHIP_FAST_FWD 16u            /* forward bytes (beyond the 4 compared) a lane resolves on its own; this many = "or more" */
#define ZHIP_WIN_EXT_NEED 84u        /* a window that resolves matches in the lanes reads up to 20 bytes beyond its last position */

template <uint32_t MLS, typename TAB>
__device__ __forceinline__ int window_batch(const uint8_t* __restrict__ src, uint32_t nm8, uint32_t hshift, const TAB& T, FastOut& out,
                                            uint32_t& ip0_, uint32_t& anchor_, uint32_t& rep1_, uint32_t& rep2_, uint32_t& nextStep,
                                            uint32_t prefixLow /* lowest valid match position (0 for a unit) */, uint32_t& carry_)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const B = ip0_;
    uint32_t const P = B + lane;
    uint32_t const carryIn = carry_;                                      // the previous window's last match ended at B+2: lanes 0, 1 belong to it
    carry_ = 0;
    if (carryIn) nextStep += 2;                                           // the scan starts at B+2
    if (out.pendLen) lits_flush(out);                                     // an earlier run's deferred store (and its spill) goes first
    // the lanes' own bytes: P-4 .. P+19 when the window may resolve matches in the lanes (ext), else P .. P+7
    bool ext = ZHIP_WIN_FAST && B >= 8u && B + ZHIP_WIN_EXT_NEED <= nm8 + 8u;
    uint32_t o0 = 0, cur32, o2, o3 = 0, o4 = 0, o5 = 0;
    if (ext) {
        Quad const a = ld128(src + (P - 4)); uint64_t const b = ld64(src + (P + 12));
        o0 = a.x; cur32 = a.y; o2 = a.z; o3 = a.w; o4 = (uint32_t)b; o5 = (uint32_t)(b >> 32);
    } else {
        uint64_t const c8 = ld64(src + P);
        cur32 = (uint32_t)c8; o2 = (uint32_t)(c8 >> 32);
    }
    uint64_t const cur8 = (uint64_t)cur32 | ((uint64_t)o2 << 32);
    // the bytes the two repcodes point at: loaded unconditionally (an invalid repcode, 0, reads the lane's own bytes) so that all the
    // loads of the front are in flight together — a load inside a branch is waited for inside the branch
    uint32_t const v1 = ld32(src + (P - rep1_)), v2 = ld32(src + (P - rep2_));       // a repcode offset never exceeds the position it is used at
    uint32_t const h = hash_pos<MLS>(cur8, hshift);

    // table gather; the slot doubles as the duplicate detector (lane id written, read back, old value restored — the
    // lanes of one hash hold the same old value)
    uint32_t const myTag = fast_tag15(cur32);
    bool tagMaybe;
    uint32_t const old = tab_get_t(T, h, B > 65536, myTag, tagMaybe);
    // candidates below position 8 cannot be read 4 bytes back (rare: only the first entries of a unit): such a window goes the exact way
    if (ext && __ballot(old != 0 && old < 8u)) ext = false;
    uint32_t cb, info = 0;                                                // info: bits 0-4 forward bytes, 5-7 backward bytes, 8 = the lane's candidate is the table's
    if (ext) {
        uint32_t const cbase = old ? old - 4 : 4u;                        // old == 0 reads harmless bytes; in flight during the duplicate detection below
        Quad const c = ld128(src + cbase); uint64_t const d = ld64(src + cbase + 16);
        cb = c.y;
        uint32_t const xb = o0 ^ c.x;
        uint32_t const back = xb ? (uint32_t)__clz((int)xb) >> 3 : 4u;
        uint32_t f = same_lo(o5 ^ (uint32_t)(d >> 32));
        {   uint32_t t;
            t = same_lo(o4 ^ (uint32_t)d); f = t + (t == 4u ? f : 0u);
            t = same_lo(o3 ^ c.w); f = t + (t == 4u ? f : 0u);
            t = same_lo(o2 ^ c.z); f = t + (t == 4u ? f : 0u); }
        info = f | (back << 5) | 0x100u;
    } else {
        // only the lanes whose tag agrees fetch their candidate's bytes (an empty slot or a different tag cannot match: zhip_parse.h, TAGS); the
        // others all read the unit's first bytes — one line, one request — so that the load stays unconditional: a load inside a branch is
        // waited for inside the branch (round 3), and this one has the duplicate detection below to hide behind
        bool const fetch = old != 0 && tagMaybe;
        cb = ld32(src + tab_guard(T, fetch ? old : 0u));
        if (!fetch) cb = ~cur32;
    }
    uint32_t backId = lane;
    if constexpr (!TabTraits<TAB>::ballotGroups) {
        __builtin_amdgcn_wave_barrier();
        tab_mark(T, h, lane);
        __builtin_amdgcn_wave_barrier();
        backId = tab_peek(T, h);
        __builtin_amdgcn_wave_barrier();
        tab_unmark(T, h, old);
        __builtin_amdgcn_wave_barrier();
    }

    // NF: lanes that share their hash with an earlier lane of the window.  What such a lane finds in the table depends on which of
    // its group's earlier members have been inserted when it is looked up: the closest inserted one, else the table's old entry.
    // Every lane keeps its two closest earlier members (p1, p2) and whether their bytes match its own (m1, m2); the event loop
    // combines them with the insert mask.  A lane with three or more earlier members (or, after ZHIP_WIN_GROUPS groups, any lane
    // from the first unresolved flagged one on) cannot be resolved: the scan stops in front of it (DEEP) — word-salad text has a repeated hash in most windows, and stopping at the
    // SECOND sharing lane (round-2 start) held its windows to 35 of 60 lanes on average.
    unsigned long long NF = 0;
    unsigned long long DEEP = lanes_from(ZHIP_WIN_LANES);      // lanes that cannot be searched from this window: >= 60, or beyond what the group data resolves
    uint32_t p1 = 0, p2 = 0, m1 = 0, m2 = 0, depth = 0;
    {   unsigned long long myG = 0;
        if constexpr (TabTraits<TAB>::ballotGroups) {
            myG = wave_hash_group(h, 32u - hshift);                       // every group exactly, whatever their number
            NF = __ballot((myG & lanes_below(lane)) != 0);
        } else {
        unsigned long long ML = __ballot(backId != lane);
        int it = 0;
        while (ML) {
            uint32_t const j = ff1u(ML);
            if (it == ZHIP_WIN_GROUPS) { NF |= lanes_from(j); DEEP |= lanes_from(j); break; }
            uint32_t const hj = __builtin_amdgcn_readlane(h, (int)j);
            unsigned long long const G = __ballot(h == hj);
            if (h == hj) myG = G;
            NF |= G & (G - 1);
            ML &= ~G; it++;
        }
        }
        if (NF) {
            unsigned long long const prev = myG & lanes_below(lane);
            depth = (uint32_t)__builtin_popcountll(prev);
            p1 = prev ? 63u - (uint32_t)__clzll((long long)prev) : 0u;
            unsigned long long const prev2 = prev & ~(1ull << p1);
            p2 = prev2 ? 63u - (uint32_t)__clzll((long long)prev2) : 0u;
            uint32_t const c1 = __shfl(cur32, (int)p1), c2 = __shfl(cur32, (int)p2);
            m1 = (depth >= 1 && c1 == cur32) ? 1u : 0u;
            m2 = (depth >= 2 && c2 == cur32) ? 1u : 0u;
            DEEP |= __ballot(depth >= 3);
        }
    }
    bool const hitOld = old != 0 && old >= prefixLow && cb == cur32;
    unsigned long long const M = __ballot(hitOld);
    uint32_t const x1 = rep1_ ? cur32 ^ v1 : 1u, x2 = rep2_ ? cur32 ^ v2 : 1u;
    uint32_t const offv = P - old;                                        // the offset of a match with the table's candidate
    uint32_t const lanesLo = lane < 32 ? (1u << lane) - 1u : ~0u, lanesHi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;   // the lanes below this one
    ZWPROF(out, 1);

    uint32_t const nbSeq0 = out.nbSeq, longPos0 = out.longPos, longType0 = out.longType;
    uint32_t const anchorEntry = anchor_;
    uint32_t const nextStep0 = nextStep;
    // a match that crosses lane 63 may hand its end to the next window only if that one is a window too
#ifdef ZHIP_DBG_NOCARRY
    uint32_t const carryMaxE = 64u;
#else
    uint32_t const carryMaxE = (B + 64u + ZHIP_FAST_FWD + 4u + ZHIP_WIN_NEED <= nm8 + 8u + 2u) ? 64u + ZHIP_FAST_FWD + 4u : 64u;   // e < this bound
#endif
    uint32_t evA = 0, evB = 0, nEv;                                       // event registers: sequence t of this window in lane t
    unsigned long long INS, COV, DIRECT;                                  // DIRECT: sequence slots of this window already stored by a chain
    uint32_t anchor, rep1, rep2, backBefore, sumLit, i;
    int status;
#define ZW_EMIT(ll, ob, ml) do { uint32_t const mb_ = (ml) - 3;                                                   \
        if (((ll) | mb_) > 0xFFFF) {                                                                              \
            if ((ll) > 0xFFFF) { out.longType = 1; out.longPos = out.nbSeq; }                                     \
            if (mb_ > 0xFFFF) { out.longType = 2; out.longPos = out.nbSeq; } }                                    \
        uint32_t const slot_ = out.nbSeq - nbSeq0;                                                                \
        evA = ZHIP_WRITELANE((ob), slot_, evA); evB = ZHIP_WRITELANE(((ll) & 0xFFFFu) | (mb_ << 16), slot_, evB); \
        out.nbSeq++; } while (0)
    // the window's table writes: the lanes of INS outside NF together, then its NF lanes one by one
#define ZW_TABLE_FLUSH() do {                                                                                    \
        unsigned long long late_ = INS & NF, CM = INS ^ late_;                                                    \
        if (__builtin_amdgcn_inverse_ballot_w64(CM)) tab_put_t(T, h, P, myTag);                                   \
        __builtin_amdgcn_wave_barrier();                                                                          \
        while (late_) { if (lane == ff1u(late_)) tab_put_t(T, h, P, myTag); late_ &= late_ - 1; __builtin_amdgcn_wave_barrier(); } \
        INS = 0; } while (0)

    // pass 0 chains events where it can; if a deferred repcode check fails, pass 1 runs the window again the exact way
    for (uint32_t pass = 0; ; pass++) {
    bool const fastOn = ext && pass == 0;
    bool failed = false;
    anchor = anchor_; rep1 = rep1_; rep2 = rep2_; nextStep = nextStep0;
    out.nbSeq = nbSeq0; out.longPos = longPos0; out.longType = longType0;
    unsigned long long E1q = __ballot(x1 == 0), E1b = __ballot((x1 & 0xFFu) == 0);
    unsigned long long E2q = __ballot(x2 == 0), E2b = __ballot((x2 & 0xFFu) == 0);
    bool e1ok = true, e2ok = true;                                       // E1* / E2* hold the masks of rep1 / rep2
    unsigned long long V = 0;                                            // lanes whose deferred repcode check is pending ...
    uint32_t vOff = 0;                                                   // ... and the offset each is compared at
    // inserts: INS collects the inserted lanes; the lanes of NF among them (only single inserts can be) are written
    // one by one after the others, in position order (a later member of a hash group overwrites an earlier one)
    INS = 0; COV = 0; DIRECT = 0; nEv = ~0u; backBefore = 0; sumLit = 0; i = 0;
    status = ZW_RESTART;
    int kLim;                                                             // iterations before the gap grows (:342-346), entry scan only
    // exact masks for both repcodes + the verdict on every deferred check, in one round of loads; false: a deferred check failed
#define ZW_RESOLVE() do {                                                                                        \
        uint32_t xv_ = 1, xa_ = 1, xb_ = 1;                                                                       \
        if (__builtin_amdgcn_inverse_ballot_w64(V)) xv_ = cur32 ^ ld32(src + (P - vOff));                          \
        if (!e1ok && rep1 && P >= rep1) xa_ = cur32 ^ ld32(src + (P - rep1));                                      \
        if (!e2ok && rep2 && P >= rep2) xb_ = cur32 ^ ld32(src + (P - rep2));                                      \
        if (__ballot(xv_ == 0) & V) failed = true;                                                                 \
        if (!e1ok) { E1q = __ballot(xa_ == 0); E1b = __ballot((xa_ & 0xFFu) == 0); e1ok = true; }                  \
        if (!e2ok) { E2q = __ballot(xb_ == 0); E2b = __ballot((xb_ & 0xFFu) == 0); e2ok = true; }                  \
        V = 0; } while (0)
    // :410-420 at lane e_ with exact masks; leaves e_ at the end of the last immediate repcode
#define ZW_IMMEDIATE(e_) do {                                                                                    \
        uint32_t const rl = 4 + fwd_run(src, nm8, B, E2b, (e_) + 4, rep2);                                         \
        {   uint32_t const t = rep2; rep2 = rep1; rep1 = t; }                                                      \
        {   unsigned long long t = E2q; E2q = E1q; E1q = t; t = E2b; E2b = E1b; E1b = t; }                         \
        INS |= 1ull << (e_);                                                                                       \
        ZW_EMIT(0u, 1u, rl);                                                                                       \
        uint32_t const en = (e_) + rl;                                                                             \
        COV |= en < 64 ? (lanes_from(e_) & lanes_below(en)) : lanes_from(e_);                                      \
        (e_) = en; anchor = B + (e_); } while ((e_) < 64 && ((E2q >> (e_)) & 1))
    // the repcode loop went on beyond the window: continue it by loads (the table must be up to date for its inserts)
#define ZW_POST_FAR(cur0_, first_) do {                                                                          \
        nEv = out.nbSeq - nbSeq0;                                                                                  \
        ZW_TABLE_FLUSH();                                                                                          \
        uint32_t ip0n = anchor;                                                                                    \
        if (ip0n <= nm8) {                                                                                         \
            PostState ps; ps.ip0 = ip0n; ps.anchor = anchor; ps.rep1 = rep1; ps.rep2 = rep2; ps.nbSeq = out.nbSeq; ps.longPos = out.longPos; ps.longType = out.longType; \
            ps = post_match_far<MLS, TAB>(src, nm8, hshift, T, out.seqs, ps, (cur0_), (first_));                  \
            ip0n = ps.ip0; anchor = ps.anchor; rep1 = ps.rep1; rep2 = ps.rep2; out.nbSeq = ps.nbSeq; out.longPos = ps.longPos; out.longType = ps.longType; \
        }                                                                                                          \
        i = ip0n - B; } while (0)
    {   int32_t const d = (int32_t)(nextStep - B) - 4;
        kLim = (d <= 0 ? 0 : (d + 1) >> 1) + 1; }
    if (carryIn) {
        // lanes 0 and 1 are the last two bytes of the previous window's last match: its second complementary insert (:408) is lane 0,
        // its immediate-repcode check (:410) is lane 2 — with exact masks, the front loaded them for the repcodes that match left
        INS = 1ull; i = 2;                                           // (lanes 0, 1 are not literals: masked out at the end, they lie in front of the anchor)
        if (rep2 && ((E2q >> 2) & 1)) {
            uint32_t e = 2;
            ZW_IMMEDIATE(e);
            i = e; nextStep = B + e + 128;
            if (e >= 64) { ZW_POST_FAR(0, false); goto window_done; }
        }
        {   int32_t const d = (int32_t)(nextStep - (B + i)) - 4;           // a fresh scan: 64 iterations before the gap grows
            kLim = (d <= 0 ? 0 : (d + 1) >> 1) + 1; }
    }
    for (;;) {
        // a lane with three or more earlier group members only matters if the scan has to look it up: inside a match (a run of equal
        // bytes is one hash group) it never is, so the bound is taken from the scan position on, every time
        uint32_t const Dw = ff1u(DEEP & lanes_from(i));                   // DEEP holds lanes 60..63, i < 64 here
        int Kw = ((int)Dw - (int)i) >> 1;
        if (Kw > kLim) Kw = kLim;
        if (Kw <= 0) break;                                               // ZW_RESTART
        uint32_t const hiLane = i + 2u * (uint32_t)Kw;                    // searched lanes i .. hiLane-1 (<= 60), probes up to hiLane
        unsigned long long const span = ZHIP_SBFM64(hiLane - i, i);
        unsigned long long Me = M;
        uint32_t candSel = old;
        uint32_t infoSel = info;
        if (NF) {
            // a member of a hash group is inserted when the scan from i reaches it before the lane in question (every lane from i on
            // is, as long as no event intervenes — and the first event is what is being looked for), or when INS already holds it
            unsigned long long const insE = INS | lanes_from(i);
            bool const in1 = depth >= 1 && ((insE >> p1) & 1), in2 = depth >= 2 && ((insE >> p2) & 1);
            bool const hit = in1 ? (m1 != 0) : (in2 ? (m2 != 0) : hitOld);
            candSel = in1 ? B + p1 : (in2 ? B + p2 : old);
            if (in1 || in2) infoSel = 0;                                  // a candidate inside the window: not resolved in the lane
            Me = __ballot(hit);
        }
        unsigned long long const MM = Me & span;
        unsigned long long const parity = (i & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
        // the repcode probes of this scan: exact when E1* is (after a chain it is not: the events that follow get exact masks first)
        unsigned long long const RP = e1ok ? (E1q & parity & (span << 2)) : 0ull;
        uint32_t const jm = ff1u(MM), jr = ff1u(RP);
        if (!e1ok && rep1 && (int32_t)jm >= 0) {
            // an event after a chain: settle the chain's deferred checks and this scan's own probes, get exact masks, look again
            unsigned long long const VM = parity & ZHIP_SBFM64(((jm - i) & ~1u) + 1u, i + 2);
            if (__builtin_amdgcn_inverse_ballot_w64(VM)) vOff = rep1;
            V |= VM;
            ZWPROF_COUNT(out, 14, 1);
            ZW_RESOLVE();
            if (failed) break;
            // all probes up to the match were just verified negative: the match at jm stands (isRep below is 0: jr is empty)
        }
        if ((int32_t)(jm & jr) < 0) {                                     // neither
            if (!e1ok && rep1) {                                          // the tail of a chain: its probes join the deferred checks
                unsigned long long const VM = parity & (span << 2);
                if (__builtin_amdgcn_inverse_ballot_w64(VM)) vOff = rep1;
                V |= VM;
            }
            INS |= span;
            i = hiLane;
            status = (Kw == kLim) ? ZW_INC : ZW_CONT;
            break;
        }
        ZWPROF(out, 2);
        ZWPROF_COUNT(out, 11, 1);
        // the repcode probe of an iteration comes before its two matches (:268-290, then :292-299 / :317-326)
        uint32_t const isRep = ((jr - i - 2) >> 1) <= ((jm - i) >> 1) ? 1u : 0u;
        uint32_t const j = isRep ? jr : jm;
        uint32_t const c = __builtin_amdgcn_readlane(candSel, (int)j);
        uint32_t const room = B + j - anchor;
        uint32_t const limit = isRep ? 1u : (room < c - prefixLow ? room : c - prefixLow);        // :271 / :387 (match0 > prefixStart)
        if (fastOn && !isRep && e1ok && room <= 0xFFFFu) {
            uint32_t const inf0 = __builtin_amdgcn_readlane(infoSel, (int)j);
            uint32_t const fl0 = inf0 & 31u, braw0 = (inf0 >> 5) & 7u;
            uint32_t const e0 = j + 4 + fl0;
            bool const immHit = rep1 != 0 && e0 < 64u && ((E1q >> e0) & 1);       // the immediate repcode after this match would hit (rep2 = today's rep1)
            if ((inf0 & 0x100u) && fl0 < ZHIP_FAST_FWD && (braw0 < 4u || limit <= 4u) && e0 < carryMaxE && !immHit) {
                // ================= a CHAIN of events starts at lane j =================
                // ---- the chase: J collects the event lanes, s is the scan position behind the last one
                unsigned long long J = 1ull << j;
                uint32_t s = e0, jLast = j, hiT = 0;
                int mode = 0;                                             // how the chain ends: 0 at s (nothing searchable / an event for the exact code), 1 = scan ran out (tail to hiT), 2 = crossing
                unsigned long long insSoFar = INS | ZHIP_SBFM64(j + 3 - i, i) | (1ull << ((e0 - 2) & 63));   // what the NF lanes ahead need to know
                for (;;) {
                    if (s >= 64u) { mode = 2; break; }
#ifdef ZHIP_DBG_CHAIN1
                    break;
#endif
                    uint32_t const Dc = ff1u(DEEP & lanes_from(s));
                    int const Kc = ((int)Dc - (int)s) >> 1;
                    if (Kc <= 0) break;
                    uint32_t const hiC = s + 2u * (uint32_t)Kc;
                    unsigned long long const spanC = ZHIP_SBFM64(hiC - s, s);
                    unsigned long long MeC = M;
                    uint32_t infoC = info;
                    if (NF & spanC) {
                        unsigned long long const insE = insSoFar | lanes_from(s);
                        bool const in1 = depth >= 1 && ((insE >> p1) & 1), in2 = depth >= 2 && ((insE >> p2) & 1);
                        bool const hit = in1 ? (m1 != 0) : (in2 ? (m2 != 0) : hitOld);
                        if (in1 || in2) infoC = 0;
                        MeC = __ballot(hit);
                    }
                    uint32_t const jn = ff1u(MeC & spanC);
                    if ((int32_t)jn < 0) { mode = 1; hiT = hiC; break; }
                    uint32_t const infn = __builtin_amdgcn_readlane(infoC, (int)jn);
                    uint32_t const fln = infn & 31u, en = jn + 4 + fln;
                    if (!((infn & 0x100u) && fln < ZHIP_FAST_FWD && (((infn >> 5) & 7u) < 4u || jn - s <= 4u) && en < carryMaxE)) break;
                    J |= 1ull << jn;
                    insSoFar |= ZHIP_SBFM64(jn + 3 - s, s) | (1ull << ((en - 2) & 63));
                    jLast = jn; s = en;
                }
                bool trim = false;
                if (mode == 1 && J == (1ull << j) && hiT - s < ZHIP_FAST_TAIL) { mode = 0; trim = true; ZWPROF_COUNT(out, 15, 1); }   // one event, short tail: leave the tail to the next window
                // ---- every event of the chain in its own lane (straight-line code: flags are 0/1 words, no branches)
                uint32_t const Jlo = (uint32_t)J, Jhi = (uint32_t)(J >> 32);
                uint32_t const selfLo = lane < 32 ? 1u << lane : 0u, selfHi = lane < 32 ? 0u : 1u << (lane - 32);
                uint32_t const isEv = ((Jlo & selfLo) | (Jhi & selfHi)) ? 1u : 0u;
                uint32_t const bLo = Jlo & lanesLo, bHi = Jhi & lanesHi;             // events below this lane
                uint32_t const hasP = (bLo | bHi) ? 1u : 0u;
                uint32_t const jp = hasP ? top_bit(bLo, bHi) : 0u;
                uint32_t const rank = (uint32_t)__popc(bLo) + (uint32_t)__popc(bHi);
                uint32_t const flL = info & 31u, brawL = (info >> 5) & 7u;
                uint32_t const endL = lane + 4 + flL;                                // where my match would end
                uint32_t const endP = pull(endL, jp), offP = pull(offv, jp);         // round 1: the event before this lane
                int32_t const aPrev = hasP ? (int32_t)endP : (int32_t)(anchor - B);  // where the literals in front of my match start
                uint32_t const sPrev = hasP ? endP : i;                              // where the scan that found me started
                uint32_t const roomL = (uint32_t)((int32_t)lane - aPrev);
                uint32_t const limL = umin32(roomL, old - prefixLow);
                uint32_t const backL = umin32(brawL, limL);
                int32_t const startL = (int32_t)lane - (int32_t)backL;
                if (isEv) {
                    ZhipSeq q; q.offBase = offv + 3; q.litLength = (uint16_t)(roomL - backL); q.mlBase = (uint16_t)(1 + backL + flL);
                    out.seqs[out.nbSeq + rank] = q;
                }
                // round 2: the start of the next event (its backward extension takes literals away), and what the event before me
                // knew about ITS predecessor (offset | parity of its scan start << 31) for the deferred checks that use that offset
                uint32_t const aLo = Jlo & ~lanesLo & ~selfLo, aHi = Jhi & ~lanesHi & ~selfHi;
                uint32_t const hasN = (aLo | aHi) ? 1u : 0u;
                uint32_t const jn2 = hasN ? low_bit(aLo, aHi) : 0u;
                int32_t const startN = (int32_t)pull((uint32_t)startL, jn2);
                uint32_t const kn = (hasP ? offP : 0u) | (sPrev << 31);              // (first event of the chain: no deferred check uses it)
                uint32_t const knP = pull(kn, jp);
                uint32_t const offPP = knP & 0x7FFFFFFFu;                            // the offset of the event before my predecessor
                // owner of this lane = the last event at or below it
                uint32_t const hasW = isEv | hasP;
                uint32_t const wL = isEv ? lane : jp, endW = isEv ? endL : endP;
                uint32_t const inSpan = hasW & (lane < endW ? 1u : 0u);
                uint32_t const cov = inSpan | (hasN & ((int32_t)lane >= startN ? 1u : 0u));
                uint32_t const tail1 = mode == 1 ? 1u : 0u;
                uint32_t const lastX = (mode == 2 && wL == jLast) ? 1u : 0u;          // my owner is a crossing match: the next window inserts its end - 2
                uint32_t const insIn = (lane <= wL + 2 ? 1u : 0u) | ((lane + 2 == endW ? 1u : 0u) & (lastX ^ 1u));
                uint32_t const insOut = (lane >= (hasW ? endW : i) ? 1u : 0u) & (hasN | (tail1 & (lane < hiT ? 1u : 0u)));
                uint32_t const ins = inSpan ? insIn : insOut;
                // deferred checks.  Probes of the scan behind event w (= jp for a lane that is no event): lanes endW+2, +4 .. up to the lane of
                // the next event (which, an event lane itself, takes the same rule), and up to hiT behind the last event; offset = w's.
                // Probe inside the span of event w, by the scan that FOUND w: lane w+2 or w+1 (parity of that scan's start), offset = the
                // event before w's; not for the chain's first event (exact).  Immediate check where event w ended: offset = the event
                // before w's; not for the chain's first event either (checked above).
                uint32_t const pFirst = jp == j ? 1u : 0u;                           // my predecessor is the chain's first event
                uint32_t const even = ((lane - endP) & 1u) ^ 1u;                     // same parity as the scan that started where my predecessor ended
                uint32_t const behind = hasP & (lane >= endP + 2 ? 1u : 0u) & even & (isEv | hasN | (tail1 & (lane <= hiT ? 1u : 0u)));
                uint32_t const atEnd = hasP & (lane == endP ? 1u : 0u) & (pFirst ^ 1u);
                uint32_t const inside = (isEv ^ 1u) & hasP & (lane < endP ? 1u : 0u) & (pFirst ^ 1u) & (lane == jp + 2 - ((jp - (knP >> 31)) & 1u) ? 1u : 0u);
                uint32_t const role = behind | atEnd | inside;
                uint32_t const myOff = behind ? offP : offPP;
                unsigned long long const VMc = __ballot(role && myOff != 0);
                vOff = role ? myOff : vOff;
                V |= VMc;
                // ---- back to the wave-uniform state
                uint32_t const cnt = (uint32_t)__builtin_popcountll(J);
                unsigned long long const covM = __ballot(cov), insM = __ballot(ins);
                int32_t const start0 = (int32_t)__builtin_amdgcn_readlane((uint32_t)startL, (int)j);
                int32_t const startZ = (int32_t)__builtin_amdgcn_readlane((uint32_t)startL, (int)jLast);
                uint32_t const endZ = __builtin_amdgcn_readlane(endL, (int)jLast);
                uint32_t const offZ = __builtin_amdgcn_readlane(offv, (int)jLast);
                uint32_t const offY = cnt >= 2 ? __builtin_amdgcn_readlane(offP, (int)jLast) : rep1;
                uint32_t const bb = start0 < 0 ? (uint32_t)(-start0) : 0u;
                if (bb) backBefore = bb;
                // literals the chain's sequences carry = everything between the entry anchor and the last match's start that no match covers
                {   uint32_t const upto = startZ > 0 ? (uint32_t)startZ : 0u;
                    unsigned long long below = upto < 64 ? lanes_below(upto) : ~0ull;
                    if ((int32_t)(anchorEntry - B) > 0) below &= lanes_from(anchorEntry - B);             // a carried window: lanes 0, 1 lie in front of the anchor
                    uint32_t const covered = (uint32_t)__builtin_popcountll((COV | covM) & below) + (startZ >= 0 ? backBefore : 0u);   // (bytes a match took back in front of lane 0)
                    uint32_t const total = (uint32_t)(startZ - (int32_t)(anchorEntry - B));       // bytes from the entry anchor to the last match's start
                    sumLit = total - covered; }
                DIRECT |= ZHIP_SBFM64(cnt, out.nbSeq - nbSeq0);
                out.nbSeq += cnt;
                COV |= covM; INS |= insM;
                rep2 = offY; rep1 = offZ; e1ok = false; e2ok = (cnt == 1);
                if (cnt == 1) { E2q = E1q; E2b = E1b; }
                anchor = B + endZ;
                ZWPROF_COUNT(out, 12, cnt);
#ifdef ZHIP_DBG_PRINT
                if (lane == 0) printf("chain B=%u i=%u J=%llx mode=%d trim=%d s=%u jLast=%u endZ=%u startZ=%d off=%u/%u V=%llx INS=%llx COV=%llx nbSeq=%u sumLit=%u\n", B, i, J, mode, (int)trim, s, jLast, endZ, startZ, offZ, offY, V, INS, COV, out.nbSeq, sumLit);
#endif
                if (mode == 2) {                                          // the last match crossed lane 63: the next window starts at its end - 2
                    i = endZ - 2; carry_ = 1;
                    break;
                }
                kLim = 64; nextStep = B + endZ + 128;                     // _start: a fresh scan behind the last match
                if (mode == 1) { i = hiT; status = ZW_CONT; break; }
                i = endZ;
                if (trim) break;                                          // ZW_RESTART at the end of the match
                ZWPROF(out, 4);
                continue;
            }
        }
        // ---- exact event
        uint32_t const off = isRep ? rep1 : B + j - c;
        // lanes i .. j are inserted (hash0 = ip0); a repcode is found at ip2 = j, i.e. up to j-1 = ip1 (:283); a match
        // also inserts ip1 = j+1 (:296, :323 step <= 4)
        INS |= ZHIP_SBFM64(j + 1 - isRep - i, i);
        if (!isRep) {
            INS |= 1ull << (j + 1);
            rep2 = rep1; E2q = E1q; E2b = E1b; rep1 = off;
            uint32_t x = 1;
            if (P >= off) x = cur32 ^ ld32(src + (P - off));
            E1q = __ballot(x == 0); E1b = __ballot((x & 0xFFu) == 0);
        }
        uint32_t run = 0;
        if (j) {
            unsigned long long const t = ~E1b << (64 - j);
            run = t ? (uint32_t)__clzll((long long)t) : j;
        }
        if (run == j && limit > run) run += wave_count_back(src, B, B - off, limit - run);
        uint32_t const back = run < limit ? run : limit;
        uint32_t const fl = fwd_run(src, nm8, B, E1b, j + 4, off);
        uint32_t const mLength = 4 + back + fl;
        ZWPROF(out, 3);
        uint32_t const ll = room - back;
        ZW_EMIT(ll, isRep ? 1u : off + 3, mLength);
        sumLit += ll;
        uint32_t sL = j - back;
        if (back > j) { backBefore = back - j; sL = 0; }
        uint32_t e = j + 4 + fl;
        anchor = B + e;
        uint32_t const cur0L = j - 2 * isRep;
        if (e >= 64) {                                                    // ran past the window: :403-420 by loads
            COV |= lanes_from(sL);
            ZW_POST_FAR(B + cur0L, true);
            ZWPROF(out, 9);
            break;
        }
        COV |= ZHIP_SBFM64(e - sL, sL);
        INS |= (1ull << (cur0L + 2)) | (1ull << (e - 2));                 // :407-408 (ip0 <= ilimit inside a window)
        if (rep2 && ((E2q >> e) & 1)) {                                   // :410-420
            ZW_IMMEDIATE(e);
            if (e >= 64) { ZW_POST_FAR(0, false); break; }
        }
        i = e;
        kLim = 64; nextStep = B + e + 128;                                // _start: a fresh scan inside the window
        ZWPROF(out, 4);
    }
window_done:
    // deferred checks still open at the end of the scan (nothing of the window has been written yet but sequence records)
    if (!failed && V) {
        ZWPROF_COUNT(out, 14, 1);
        uint32_t xv = 1;
        if (__builtin_amdgcn_inverse_ballot_w64(V)) xv = cur32 ^ ld32(src + (P - vOff));
        if (__ballot(xv == 0) & V) failed = true;
        V = 0;
    }
#undef ZW_RESOLVE
#undef ZW_IMMEDIATE
#undef ZW_POST_FAR
    if (!failed) break;
    carry_ = 0;
    ZWPROF_COUNT(out, 13, 1);
    }
    ZWPROF(out, 2);
#undef ZW_EMIT
#ifdef ZHIP_DBG_PRINT
    if (lane == 0) printf("  window B=%u carry=%u ext=%d -> i=%u status=%d INS=%llx NF=%llx COV=%llx nbSeq=%u rep=%u/%u anchor=%u carryOut=%u\n", B, carryIn, (int)ext, i, status, INS, NF, COV, out.nbSeq, rep1, rep2, anchor, carry_);
#endif
    ZW_TABLE_FLUSH();
#undef ZW_TABLE_FLUSH
    // its sequences (those of a chain are in place already)
    if (nEv == ~0u) nEv = out.nbSeq - nbSeq0;
    if (lane < nEv && !((DIRECT >> lane) & 1)) {
        ZhipSeq q; q.offBase = evA; q.litLength = (uint16_t)evB; q.mlBase = (uint16_t)(evB >> 16);
        out.seqs[nbSeq0 + lane] = q;
    }
    {   // its literals: the lanes behind the new scan position that no match covers (the tail after the last match is
        // tentative: a later backward extension may take it back, its bytes are then simply overwritten)
        unsigned long long LIT = i < 64 ? (~COV & lanes_below(i)) : ~COV;
        if (carryIn) LIT &= ~3ull;
        if (__builtin_amdgcn_inverse_ballot_w64(LIT)) {
            uint32_t const before = __builtin_amdgcn_mbcnt_hi((uint32_t)(COV >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)COV, 0));
            out.lits[out.litPos + (B - anchorEntry) + lane - before - backBefore] = (uint8_t)cur8;
        }
    }
    out.litPos += sumLit;
    ip0_ = B + i; anchor_ = anchor; rep1_ = rep1; rep2_ = rep2;
    ZWPROF(out, 5);
    return status;
}

// One block of ZSTD_fast over src[b0, n) with the table T as the previous blocks of the same frame left it (a unit: b0 = 0 and
// a fresh table).  Matches may start anywhere in [prefixLow, position): prefixLow = max(0, n - 2^windowLog) is where
// ZSTD_window_enforceMaxDist (zstd_compress.c:4686) leaves the window for this block; maxRep = ip0 - windowLow (:238-244).
template <uint32_t MLS, typename TAB>
__device__ inline void parse_fast_block(const uint8_t* __restrict__ src, uint32_t b0, uint32_t n, uint32_t prefixLow, uint32_t maxRep,
                                        uint32_t repIn1, uint32_t repIn2, uint32_t repIn3, const ZhipUnit& u,
                                        const TAB& T, ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const hlog = u.hashLog, hshift = 32 - hlog;
    uint32_t const stepSize = u.targetLength + !u.targetLength + 1;         // zstd_fast.c:200
    FastOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.longPos = 0; out.longType = 0;
    out.litPos = 0; out.pendV = 0; out.pendSh = 0; out.pendOff = 0; out.pendLen = 0;
    ZPROF_DECL
#ifdef ZHIP_PROF
    out.zp = zp_acc_; out.zlast = &zp_last_;
#endif
    ZPROF(0);

    uint32_t anchor = b0, rep1 = repIn1, rep2 = repIn2, saved1 = 0, saved2 = 0;
    // :238-244  a repcode that reaches below the window is set aside for the block
    if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
    if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }

    if (n - b0 >= 12) {                     // shorter blocks never run an iteration (ip3 = ip0 + 2 + 1 < n - 8); keeps n - 8 >= b0
    uint32_t const nm8 = n - 8;
    int32_t const ilimit = (int32_t)nm8;
    uint32_t ip0 = b0 + (b0 == prefixLow);                                   // :238 ip0 += (ip0 == prefixStart)
    if (ip0 != b0 && lane == 0) lits[0] = src[b0];                           // that position is never searched: windows only store their own lanes
    uint32_t startPosOff, startRposOff; batch_offsets(stepSize, stepSize, startPosOff, startRposOff);
    unsigned long long const evenLanes = 0x5555555555555555ull;

    uint32_t carry = 0;                     // a window handed the end of its last match to the next one (see window_batch)
    bool have = false;                      // `cur` already holds the bytes of the batch that starts at ip0
    FastBatch cur; cur.bytes = 0; cur.rcur = 0; cur.rv = 0;
    for (;;) {                                                               // one turn per `_start`
        uint32_t step = stepSize, g0 = stepSize, nextStep = ip0 + 128;
        if ((int32_t)(ip0 + g0 + 1) >= ilimit) break;                        // :257
        uint32_t posOff = startPosOff, rposOff = startRposOff;

        // ---- scan until an event or the end of the unit: windows while the gap is 2, schedule-shaped batches otherwise
        int evKind = 0;                      // 0 none (unit finished), 1 match, 2 repcode, 3 the window handled its events
        uint32_t mpos = 0, cand0 = 0, cur0 = 0;
        bool dense = ZHIP_WIN_DENSE_ONLY == 0;   // windows right behind an event; a scan that found nothing in a whole window goes on in
        dense = true;                            // schedule-shaped batches (cheaper per position when events are far apart)
        for (;;) {
            if (dense && stepSize == 2 && step == 2 && g0 == 2 && ip0 + ZHIP_WIN_NEED <= n) {
                int const st = window_batch<MLS, TAB>(src, nm8, hshift, T, out, ip0, anchor, rep1, rep2, nextStep, prefixLow, carry);
                have = false;
                ZPROF_COUNT(10, 1);
                if (st == ZW_RESTART) { evKind = 3; break; }
                if (st == ZW_INC) { step = 3; nextStep += 128; batch_offsets(g0, step, posOff, rposOff); }
                if (ZHIP_WIN_DENSE_ONLY) dense = false;
                continue;
            }
            if (!have) cur = batch_load(src, nm8, ip0, posOff, rposOff, rep1);
            have = false;
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
            uint32_t const myTag = fast_tag15(cur32);
            bool tagMaybe;
            uint32_t const old = tab_get_t(T, h, ip0 > 65536, myTag, tagMaybe);
            uint32_t back = lane;
            if constexpr (!TabTraits<TAB>::ballotGroups) {
                __builtin_amdgcn_wave_barrier();
                if (live) tab_mark(T, h, lane);
                __builtin_amdgcn_wave_barrier();
                back = tab_peek(T, h);
            }

            // speculative loads for the next batch (used if this one has no event)
            uint32_t const nip0 = ip0 + g0 + (uint32_t)(K - 1) * step;
            uint32_t const nstep = step + (uint32_t)(K - 1 == kInc);
            uint32_t nposOff = posOff, nrposOff = rposOff;
            if (g0 != step || nstep != step) batch_offsets(step, nstep, nposOff, nrposOff);
            FastBatch const nxt = batch_load(src, nm8, nip0, nposOff, nrposOff, rep1);

            bool const fetch = old != 0 && tagMaybe;                        // an empty slot or a different tag cannot match: those lanes share one harmless line
            uint32_t cb = ld32(src + tab_guard(T, fetch ? old : 0u));
            if (!fetch) cb = ~cur32;
            uint32_t cand = old;
            unsigned long long dupMask, grp = 0;
            if constexpr (TabTraits<TAB>::ballotGroups) {
                grp = wave_hash_group(h, 32u - hshift) & liveMask;
                dupMask = __ballot(live && (grp & (grp - 1)) != 0);
                if (!live || !(grp & (grp - 1))) grp = 0;                    // (lanes outside a group carry no group, like the loop below leaves them)
            } else dupMask = __ballot(back != lane) & liveMask;
            if (dupMask) {
                // exact groups of live lanes with equal hash; a lane's candidate is its closest earlier group member
                if constexpr (!TabTraits<TAB>::ballotGroups) {
                unsigned long long ML = dupMask;
                while (ML) {
                    int const j = first_lane(ML);
                    uint32_t const hj = __builtin_amdgcn_readlane(h, j);
                    unsigned long long const G = __ballot(h == hj) & liveMask;
                    if (h == hj) grp = G;
                    ML &= ~G;
                }
                }
                unsigned long long const prevMask = grp & below_mask((int)lane);
                uint32_t const pd = prevMask ? 63u - (uint32_t)__clzll((long long)prevMask) : lane;
                uint32_t const dpos = __shfl(pos, (int)pd), d32 = __shfl(cur32, (int)pd);
                if (prevMask) { cand = dpos; cb = d32; }
            }
            unsigned long long const mMask = __ballot(cand != 0 && cand >= prefixLow && cb == cur32) & liveMask;
            unsigned long long const rMask = rep1 ? (__ballot(cur.rcur == cur.rv) & liveMask & evenLanes) : 0ull;
            int const jm = mMask ? first_lane(mMask) : 64, jr = rMask ? first_lane(rMask) : 64;
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
            if (we) { if (inC) tab_put_t(T, h, pos, myTag); else tab_unmark(T, h, old); }
            __builtin_amdgcn_wave_barrier();
            ZPROF(6);

            if (evKind == 1) {
                mpos = __builtin_amdgcn_readlane(pos, jm);
                cand0 = __builtin_amdgcn_readlane(cand, jm);
                cur0 = mpos;
                if ((jm & 1) && step <= 4) {                                 // :318-324 hashTable[hash1] = ip1 (= A_{k+1})
                    // A_{k+1} is lane jm+1's position (its hash is at hand); for jm == 63 it opens the next batch
                    if (jm < 63) { if ((int)lane == jm + 1) tab_put_t(T, h, pos, myTag); }
                    else if (lane == 0) tab_put_t(T, hash_pos<MLS>(nxt.bytes, hshift), nip0, fast_tag15((uint32_t)nxt.bytes));
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
            cur = nxt; have = true;
        }
        if (evKind == 0) break;
        if (evKind == 3) continue;                                           // the window left ip0 at the next `_start`

        // ---- _offset / _match (:377-401)
        uint32_t offBase, lim;
        if (evKind == 1) {
            rep2 = rep1; rep1 = mpos - cand0;
            offBase = rep1 + 3;
            lim = (mpos - anchor) < cand0 - prefixLow ? (mpos - anchor) : cand0 - prefixLow;       // :387 match0 > prefixStart
        } else {
            cand0 = mpos - rep1;
            offBase = 1;
            lim = 1;                                                         // :271 mLength = ip0[-1] == match0[-1]
        }
        uint32_t backLen, fwdLen;
        ZPROF(6);
        wave_extend(src, nm8, mpos, cand0, lim, backLen, fwdLen);
        ip0 = mpos - backLen;
        {   uint32_t const mLength = 4 + backLen + fwdLen;
            lits_copy(out, src, nm8, anchor, ip0 - anchor);
            store_seq(out, ip0 - anchor, offBase, mLength);
            ip0 += mLength; anchor = ip0;
        }
        // ---- :403-420 complementary inserts + immediate repcode; the next batch's bytes ride along
        have = false;
        if ((int32_t)ip0 <= ilimit)
            have = post_match<MLS, true, TAB>(src, nm8, hshift, T, out, ip0, anchor, rep1, rep2, cur0, true, startPosOff, startRposOff, cur);
        ZPROF(7);
    }
    lits_copy(out, src, nm8, anchor, n - anchor);                           // trailing literals (zstd_compress.c:3365)
    lits_flush(out);
    } else {
        for (uint32_t i = lane; i < n - b0; i += 64) lits[i] = src[b0 + i]; // tiny block: everything is a literal
        out.litPos = n - b0;
    }
    // ---- _cleanup (:368-375)
    ZPROF(8);
    ZPROF_FLUSH(0);
    saved2 = (saved1 != 0 && rep1 != 0) ? saved1 : saved2;
    if (lane == 0) {
        meta->nbSeq = out.nbSeq; meta->lastLits = n - anchor;
        meta->longPos = out.longPos; meta->longType = out.longType;
        meta->rep[0] = rep1 ? rep1 : saved1; meta->rep[1] = rep2 ? rep2 : saved2; meta->rep[2] = repIn3;
        meta->status = 0; meta->litSize = out.litPos; meta->pad0 = 0;
    }
}

// One unit = one block with a fresh table.  smem: fast_tag_lds_bytes(hashLog) bytes of wave-private LDS
template <uint32_t MLS>
__device__ __forceinline__ void parse_fast_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u,
                                       unsigned char* smem, ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const hlog = u.hashLog;
    FastTagTab T;
    T.lo = (lds_u16*)(uintptr_t)smem;
    T.hi = (lds_u32*)(uintptr_t)(smem + (2u << hlog));
    T.tg = (lds_u32*)(uintptr_t)(smem + fast_lds_bytes(hlog));
    {   // fresh table (zstd_compress.c:2020): lo[], hi[] and the tag plane are contiguous
        lds_u32* const z = (lds_u32*)(uintptr_t)smem;
        uint32_t const words = fast_tag_lds_bytes(hlog) >> 2;
        for (uint32_t i = lane; i < words; i += 64) z[i] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    parse_fast_block<MLS, FastTagTab>(src, 0, n, 0, 1, 1, 4, 8, u, T, seqs, lits, meta);   // lowest index 0, ip0 = 1 -> maxRep = 1
}

// The same unit on a table in global memory (`gtab`: 1 << hashLog words owned by this wavefront, reused from unit to unit)
template <uint32_t MLS>
__device__ __forceinline__ void parse_fast_unit_g(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u,
                                         uint32_t* gtab, ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    uint32_t const lane = (uint32_t)lane_id();
    GlobTab T; T.w = gtab;
    {   uint32_t const words = 1u << u.hashLog;                               // fresh table (hashLog >= 6: at least one word per lane)
        for (uint32_t i = lane * 4; i < words; i += 256) { gtab[i] = 0; gtab[i + 1] = 0; gtab[i + 2] = 0; gtab[i + 3] = 0; }
    }
    __builtin_amdgcn_wave_barrier();
    parse_fast_block<MLS, GlobTab>(src, 0, n, 0, 1, 1, 4, 8, u, T, seqs, lits, meta);
}

}  // namespace zhip
