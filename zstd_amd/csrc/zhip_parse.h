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
    uint64_t* zp; uint32_t* ws; uint32_t* wh; uint64_t* zlast;   // the caller's phase accumulators (measurement build only; ws / wh: the ZSTD_fast window's)
#endif
};
// WINDOW PHASES (measurement build only, -DZHIP_PROF; scripts/isa_phase_table.py + scripts/prof_phases.py).  ZWPH(o, id) closes phase `id`: the
// s_memtime ticks since the previous marker and one visit go to the phase's slots, and a "; ZWPH id" comment lands in the assembly, from which
// the table script counts the instructions of every phase (basic blocks inherit the phase their predecessors end in).  Nothing is waited for at
// a marker: a stall is charged to the phase whose instruction stalls.  The product build expands all of it to nothing.
enum { WPH_F_SRC = 0,      // window: pending literal store, source + repcode bytes (preloaded or loaded), hash, tag
       WPH_F_TAB = 1,      // table gather (LDS: position, bit 16, tag), candidate address
       WPH_F_DUP = 2,      // candidate load issued; duplicate detection in the slots (mark / peek / unmark)
       WPH_F_GRP = 3,      // hash groups of the flagged lanes (readlane / ballot loop), p1 / p2 / m1 / m2
       WPH_F_MASK = 4,     // candidate bytes arrive: M, E1 / E2 masks (four ballots)
       WPH_SEARCH = 5,     // event loop: one search of the span (cut, NF selection, first match / first probe)
       WPH_M_ELOAD = 6,    // a match: inserts, E load of a new offset + its two ballots
       WPH_M_RUNS = 7,     // backward + forward runs from the masks (incl. the out-of-line continuation past the window)
       WPH_M_EMIT = 8,     // emit, coverage, complementary inserts, immediate-repcode loop, next scan set up
       WPH_LEAVE = 9,      // a match left the window: carry, or :403-420 by loads
       WPH_E_PRE = 10,     // window end: the next window's source bytes requested
       WPH_E_TAB = 11,     // table writes (together, then the NF lanes one by one)
       WPH_E_OUT = 12,     // sequences + literals stored
       WPH_B_SCAN = 13,    // schedule-shaped batch up to its event decision
       WPH_B_MATCH = 14,   // its match: wave_extend, literals, sequence
       WPH_B_POST = 15,    // post_match (inserts, immediate repcode, next batch's bytes)
       WPH_TAIL = 16, WPH_INIT = 17, WPH_LOOP = 18, /* between windows: the scan loop of parse_fast_block */
       WPH_CARRY = 19,     // a carried match's lanes 0..2: insert, immediate-repcode test (carry windows only)
       WPH_IMM = 20,       // one immediate-repcode loop (:410-420 from the masks)
       WPH_GRP_IT = 21,    // one hash group resolved (readlane + ballot)
       WPH_GRP_NF = 22,    // a window with groups: p1 / p2 / m1 / m2 of every lane
       WPH_LATE = 23,      // one late (NF) insert
       WPH_LEAVE_FAR = 24  /* :403-420 by loads (post_match_far) */ };
#ifdef ZHIP_PROF
#define ZWPROF(o, i) do { uint64_t const t_ = __builtin_amdgcn_s_memtime(); (o).zp[i] += t_ - *(o).zlast; *(o).zlast = t_; } while (0)      /* the lazy parser's coarse timers (zhip_parse_lazy.h) */
#define ZWPROF_COUNT(o, i, v) do { (o).zp[i] += (uint64_t)(v); } while (0)
#define ZWPROF_SYNC(o, i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ZWPROF(o, i); } while (0)   /* charges the outstanding loads to phase i */
#define ZWPH(o, i) do { asm volatile("; ZWPH %0" :: "n"(i)); uint64_t const t_ = __builtin_amdgcn_s_memtime(); (o).ws[i] += (uint32_t)(t_ - *(o).zlast); (o).wh[i]++; *(o).zlast = t_; } while (0)
#else
#define ZWPROF_SYNC(o, i) do { } while (0)
#define ZWPROF(o, i) do { } while (0)
#define ZWPROF_COUNT(o, i, v) do { } while (0)
#ifdef ZHIP_WPH_MARK                 /* the table script's build: the markers alone, on the product's code */
#define ZWPH(o, i) asm volatile("; ZWPH %0" :: "n"(i))
#else
#define ZWPH(o, i) do { } while (0)
#endif
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

// the same copy without the deferred first chunk: the ZSTD_fast parser (round 6) gives up that trick — three vector registers held through every window for the sake of the
// schedule-shaped batches' events — to fit 128 registers (a fourth wavefront per SIMD).  Like the reference's wildcopy the last 8-byte chunk of a run may spill up to 7 bytes past it.
__device__ __forceinline__ void lits_copy_now(FastOut& o, const uint8_t* src, uint32_t nm8, uint32_t from, uint32_t len)
{
    uint32_t const lane8 = 8u * (uint32_t)lane_id();
    for (uint32_t off = 0; off < len; off += 512) {
        uint32_t const q = from + off + lane8, qc = q < nm8 ? q : nm8, sh = q - qc;
        if (off + lane8 < len) st64(o.lits + o.litPos + off + lane8, ld64(src + qc) >> (8 * (sh & 7)));
    }
    o.litPos += len;
    __builtin_amdgcn_wave_barrier();        // later runs overwrite this run's spill: keep the stores in program order
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

// CARRY.  A match that runs past lane 63 hands its end to the NEXT window, which starts two positions in front of it: the complementary
// insert of end-2 (zstd_fast.c:408) is lane 0 of an ordinary window, the immediate-repcode test (:410) bit 2 of its E2 mask; the insert of
// cur0+2 (:407) is a lane of the window that found the match.  An immediate-repcode match that runs past lane 63 carries the same way
// without the insert (carry 2: the test is due at lane 0).  Only where the next scan cannot be a window does the rest of :403-420 go by
// loads (post_match_far).  (Round 3's event chains — every lane resolving its own would-be match from 24 candidate bytes, the chain's
// repcode probes checked afterwards — were byte-identical and measured slower; the carry is what is left of them.)
#ifndef ZHIP_FAST_PRELOAD
#define ZHIP_FAST_PRELOAD 1          /* measurement switch: 0 = every window loads its own source bytes */
#endif
struct FastPre { uint64_t c8; uint32_t v1, v2, B; };        // the next window's source bytes (B = ~0: none)
#ifndef ZHIP_FAST_WINNERS
#define ZHIP_FAST_WINNERS 1          /* 0: every inserted lane that shares its hash with an earlier one is written on its own, in position order (round 5) */
#endif
#ifndef ZHIP_FAST_ONE_FF1
#define ZHIP_FAST_ONE_FF1 1          /* 0: round 5's search (first match and first probe found separately, then ordered) */
#endif
#ifndef ZHIP_FAST_CARRY
#define ZHIP_FAST_CARRY 1            /* measurement switch: 0 = never carry (every leaving match goes by loads), 1 = plain matches only, 2 = immediate repcodes too */
#endif
template <uint32_t MLS, typename TAB>
__device__ __forceinline__ int window_batch(const uint8_t* __restrict__ src, uint32_t n, uint32_t nm8, uint32_t hshift, const TAB& T, FastOut& out,
                                            uint32_t& ip0_, uint32_t& anchor_, uint32_t& rep1_, uint32_t& rep2_, uint32_t& nextStep,
                                            uint32_t prefixLow /* lowest valid match position (0 for a unit) */, uint32_t& carry_, FastPre& pre)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const B = ip0_;
    uint32_t const P = B + lane;
    uint32_t const carryIn = carry_;                                      // 1: the previous window's last match ended at B+2 (lanes 0, 1 belong to it); 2: it ended at B, the immediate-repcode test is due
    carry_ = 0;
    if (carryIn == 1) nextStep += 2;                                      // the scan starts at B+2
    // the lanes' own bytes and the bytes the two repcodes point at (an invalid repcode, 0, reads the lane's own bytes): the window before this
    // one has usually requested them already (FastPre), else all three loads are in flight together
    uint64_t cur8; uint32_t v1, v2;
    if (ZHIP_FAST_PRELOAD && pre.B == B) { cur8 = pre.c8; v1 = pre.v1; v2 = pre.v2; }
    else { cur8 = ld64(src + P); v1 = ld32(src + (P - rep1_)); v2 = ld32(src + (P - rep2_)); }     // a repcode offset never exceeds the position it is used at
    pre.B = ~0u;
    uint32_t const cur32 = (uint32_t)cur8;
    uint32_t const h = hash_pos<MLS>(cur8, hshift);
    ZWPH(out, WPH_F_SRC);

    // table gather; the slot doubles as the duplicate detector (lane id written, read back, old value restored — the
    // lanes of one hash hold the same old value)
    uint32_t const myTag = fast_tag15(cur32);
    bool tagMaybe;
    uint32_t const old = tab_get_t(T, h, B > 65536, myTag, tagMaybe);
    // only the lanes whose tag agrees fetch their candidate's bytes (an empty slot or a different tag cannot match: zhip_parse.h, TAGS); the
    // others all read the unit's first bytes — one line, one request — so that the load stays unconditional: a load inside a branch is
    // waited for inside the branch (round 3), and this one has the duplicate detection below to hide behind
    bool const fetch = old != 0 && tagMaybe;
    ZWPH(out, WPH_F_TAB);
    uint32_t cb = ld32(src + tab_guard(T, fetch ? old : 0u));
    if (!fetch) cb = ~cur32;
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
    ZWPH(out, WPH_F_DUP);

    // NF: lanes that share their hash with an earlier lane of the window.  What such a lane finds in the table depends on which of
    // its group's earlier members have been inserted when it is looked up: the closest inserted one, else the table's old entry.
    // Every lane keeps its two closest earlier members (p1, p2) and whether their bytes match its own (m1, m2); the event loop
    // combines them with the insert mask.  A lane with three or more earlier members (or, after ZHIP_WIN_GROUPS groups, any lane
    // from the first unresolved flagged one on) cannot be resolved: the scan stops in front of it (DEEP) — word-salad text has a repeated hash in most windows, and stopping at the
    // SECOND sharing lane (round-2 start) held its windows to 35 of 60 lanes on average.
    unsigned long long NF = 0;
    unsigned long long DEEP = lanes_from(ZHIP_WIN_LANES);      // lanes that cannot be searched from this window: >= 60, or beyond what the group data resolves
    uint32_t grp = 0;                                          // this lane's two closest earlier group members and what they say: p1 | p2 << 6 | m1 << 12 | m2 << 13 | (depth >= 1) << 14 | (depth >= 2) << 15 (one register, held through the event loop)
    unsigned long long myG = 0;                                // the lanes of this lane's hash group (0: none, or a group the loop below did not get to)
    {
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
            ZWPH(out, WPH_GRP_IT);
        }
        }
        if (NF) {
            unsigned long long const prev = myG & lanes_below(lane);
            uint32_t const depth = (uint32_t)__builtin_popcountll(prev);
            uint32_t const p1 = prev ? 63u - (uint32_t)__clzll((long long)prev) : 0u;
            unsigned long long const prev2 = prev & ~(1ull << p1);
            uint32_t const p2 = prev2 ? 63u - (uint32_t)__clzll((long long)prev2) : 0u;
            uint32_t const c1 = __shfl(cur32, (int)p1), c2 = __shfl(cur32, (int)p2);
            uint32_t const m1 = (depth >= 1 && c1 == cur32) ? 1u : 0u;
            uint32_t const m2 = (depth >= 2 && c2 == cur32) ? 1u : 0u;
            grp = p1 | (p2 << 6) | (m1 << 12) | (m2 << 13) | ((depth >= 1 ? 1u : 0u) << 14) | ((depth >= 2 ? 1u : 0u) << 15);
            DEEP |= __ballot(depth >= 3);
            ZWPH(out, WPH_GRP_NF);
        }
    }
    ZWPH(out, WPH_F_GRP);
    bool const hitOld = old != 0 && old >= prefixLow && cb == cur32;
    unsigned long long const M = __ballot(hitOld);
    // x1 / x2: the lane's 4 bytes XOR the 4 bytes rep1 / rep2 back (1: no such repcode).  They ARE the state the event loop carries for the two repcode offsets — one
    // vector register each; the masks of round 5 (which lanes equal the byte / the 4 bytes that offset back: four 64-bit masks = eight registers in this loop, whose control
    // flow is lane-divergent for the compiler, plus four register-pair copies at every offset change) are one v_cmp away wherever they are used (ZW_Eq / ZW_Eb)
    uint32_t x1 = rep1_ ? cur32 ^ v1 : 1u, x2 = rep2_ ? cur32 ^ v2 : 1u;
#define ZW_Eq(x_) __ballot((x_) == 0)
#define ZW_Eb(x_) __ballot(((x_) & 0xFFu) == 0)

    uint32_t const anchorEntry = anchor_;
    unsigned long long INS = 0, COV = 0;
    uint32_t anchor = anchor_, rep1 = rep1_, rep2 = rep2_, backBefore = 0, sumLit = 0, i = 0;
    int status = ZW_RESTART;
#define ZW_EMIT(ll, ob, ml) do { uint32_t const mb_ = (ml) - 3;                                                   \
        if (((ll) | mb_) > 0xFFFF) {                                                                              \
            if ((ll) > 0xFFFF) { out.longType = 1; out.longPos = out.nbSeq; }                                     \
            if (mb_ > 0xFFFF) { out.longType = 2; out.longPos = out.nbSeq; } }                                    \
        /* stored as it is found (round 5 collected a window's sequences in two registers, lane t = sequence t, and stored them at its end: two registers held through the loop) */ \
        if (lane == 0) { ZhipSeq q_; q_.offBase = (ob); q_.litLength = (uint16_t)(ll); q_.mlBase = (uint16_t)mb_; out.seqs[out.nbSeq] = q_; }                                  \
        out.nbSeq++; } while (0)
    // the window's table writes.  Lanes of one hash share a slot and the reference leaves the LAST inserted one there: every lane that knows its group (myG) checks that no
    // higher member is inserted too, and all such winners write together.  Only inserted lanes of a group the front did not resolve (beyond ZHIP_WIN_GROUPS) go one by one,
    // in position order, behind the others (round 5 wrote every inserted NF lane that way: 5 200 serial writes per unit of text, profiles/r06_isa_phase_table_fast_text.txt)
#define ZW_TABLE_FLUSH() do {                                                                                    \
        unsigned long long late_ = INS & NF, CM = INS ^ late_;                                                    \
        if (ZHIP_FAST_WINNERS && late_) {                                                                         \
            unsigned long long const known_ = __ballot(myG != 0);                                                 \
            unsigned long long const win_ = __ballot((myG & INS & (lanes_from(lane) << 1)) == 0);                 \
            late_ &= ~known_;                                         /* inserted lanes of unresolved groups */   \
            CM = INS & win_ & ~late_;                                                                             \
        }                                                                                                         \
        if (__builtin_amdgcn_inverse_ballot_w64(CM)) tab_put_t(T, h, P, myTag);                                   \
        __builtin_amdgcn_wave_barrier();                                                                          \
        while (late_) { if (lane == ff1u(late_)) tab_put_t(T, h, P, myTag); late_ &= late_ - 1; __builtin_amdgcn_wave_barrier(); ZWPH(out, WPH_LATE); } \
        INS = 0; } while (0)
    ZWPH(out, WPH_F_MASK);
    // inserts: INS collects the inserted lanes; the lanes of NF among them (only single inserts can be) are written
    // one by one after the others, in position order (a later member of a hash group overwrites an earlier one)
    int kLim;                                                             // iterations before the gap grows (:342-346), entry scan only
    // :410-420 at lane e_; leaves e_ at the end of the last immediate repcode
#define ZW_IMMEDIATE(e_) do {                                                                                    \
        uint32_t const rl = 4 + fwd_run(src, nm8, B, ZW_Eb(x2), (e_) + 4, rep2);                                   \
        {   uint32_t const t = rep2; rep2 = rep1; rep1 = t; }                                                      \
        {   uint32_t const t = x2; x2 = x1; x1 = t; }                                                              \
        INS |= 1ull << (e_);                                                                                       \
        ZW_EMIT(0u, 1u, rl);                                                                                       \
        uint32_t const en = (e_) + rl;                                                                             \
        COV |= en < 64 ? (lanes_from(e_) & lanes_below(en)) : lanes_from(e_);                                      \
        (e_) = en; anchor = B + (e_); ZWPH(out, WPH_IMM); } while ((e_) < 64 && ((ZW_Eq(x2) >> (e_)) & 1))
    // the scan left the window behind a match that ended at lane e_ (>= 64; anchor = B + e_): the next window takes it over (kind_ 1: it
    // starts at the end - 2 and inserts that position, 2: at the end of an immediate repcode), or — no room for a window — :403-420 go by loads
#define ZW_LEAVE(e_, kind_, cur0_) do {                                                                          \
        if (ZHIP_FAST_CARRY >= (kind_) && B + (e_) - ((kind_) == 1 ? 2u : 0u) + ZHIP_WIN_NEED <= n) {              \
            i = (e_) - ((kind_) == 1 ? 2u : 0u); carry_ = (kind_);                                                 \
        } else {                                                                                                   \
            ZW_TABLE_FLUSH();                                                                                      \
            uint32_t ip0n = anchor;                                                                                \
            if (ip0n <= nm8) {                                                                                     \
                PostState ps; ps.ip0 = ip0n; ps.anchor = anchor; ps.rep1 = rep1; ps.rep2 = rep2; ps.nbSeq = out.nbSeq; ps.longPos = out.longPos; ps.longType = out.longType; \
                ps = post_match_far<MLS, TAB>(src, nm8, hshift, T, out.seqs, ps, (cur0_), (kind_) == 1);           \
                ip0n = ps.ip0; anchor = ps.anchor; rep1 = ps.rep1; rep2 = ps.rep2; out.nbSeq = ps.nbSeq; out.longPos = ps.longPos; out.longType = ps.longType; \
            }                                                                                                      \
            i = ip0n - B; ZWPH(out, WPH_LEAVE_FAR); } } while (0)
    {   int32_t const d = (int32_t)(nextStep - B) - 4;
        kLim = (d <= 0 ? 0 : (d + 1) >> 1) + 1; }
    if (carryIn) {
        // carry 1: lanes 0 and 1 are the last two bytes of the previous window's last match: its second complementary insert (:408) is lane 0,
        // its immediate-repcode test (:410) is lane 2; carry 2: the test is lane 0.  The front loaded exact masks for the repcodes that match left
        if (carryIn == 1) { INS = 1ull; i = 2; }                     // (lanes 0, 1 are not literals: masked out at the end, they lie in front of the anchor)
        if (rep2 && ((ZW_Eq(x2) >> i) & 1)) {
            uint32_t e = i;
            ZW_IMMEDIATE(e);
            i = e; nextStep = B + e + 128;
            if (e >= 64) { ZW_LEAVE(e, 2u, 0u); goto window_done; }
        }
        {   int32_t const d = (int32_t)(nextStep - (B + i)) - 4;           // a fresh scan: 64 iterations before the gap grows
            kLim = (d <= 0 ? 0 : (d + 1) >> 1) + 1; }
        ZWPH(out, WPH_CARRY);
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
        unsigned long long const parity = (i & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
        unsigned long long Me = M;
        uint32_t candSel = old;
        if (NF) {
            // a member of a hash group is inserted when the scan from i reaches it before the lane in question (every lane from i on
            // is, as long as no event intervenes — and the first event is what is being looked for), or when INS already holds it
            unsigned long long const insE = INS | lanes_from(i);
            uint32_t const p1 = grp & 63u, p2 = (grp >> 6) & 63u;
            bool const in1 = ((grp >> 14) & 1) && ((insE >> p1) & 1), in2 = ((grp >> 15) & 1) && ((insE >> p2) & 1);
            bool const hit = in1 ? ((grp >> 12) & 1) != 0 : (in2 ? ((grp >> 13) & 1) != 0 : hitOld);
            candSel = in1 ? B + p1 : (in2 ? B + p2 : old);
            Me = __ballot(hit);
        }
        unsigned long long const MM = Me & span;
#if ZHIP_FAST_ONE_FF1
        // the repcode probes of this scan, each moved to the FIRST lane of its iteration (the probe at ip2 = l + 2 belongs to the iteration whose ip0 is lane l): one bit scan
        // then finds the first event in the reference's order — the probe of an iteration comes before its two matches (:268-290, then :292-299 / :317-326), so at its
        // iteration's first lane it wins the tie, and the iteration's second lane carries no probe bit.  (Resolving the hash-group state lazily, for the first candidate only,
        // was measured on top of this and lost 3-6 %: profiles/r06_ab_fast_one_bit_scan_and_lazy_groups.log)
        unsigned long long const RPs = ((ZW_Eq(x1) & parity) >> 2) & span;
        unsigned long long const anyEv = MM | RPs;
        if (anyEv == 0) {                                                 // neither
            INS |= span;
            i = hiLane;
            status = (Kw == kLim) ? ZW_INC : ZW_CONT;
            break;
        }
        ZWPH(out, WPH_SEARCH);
        uint32_t const jp = ff1u(anyEv);
        uint32_t const isRep = (uint32_t)((RPs >> jp) & 1);
        uint32_t const j = jp + 2u * isRep;
#else
        unsigned long long const RP = ZW_Eq(x1) & parity & (span << 2);   // the repcode probes of this scan
        uint32_t const jm = ff1u(MM), jr = ff1u(RP);
        if ((int32_t)(jm & jr) < 0) {                                     // neither
            INS |= span;
            i = hiLane;
            status = (Kw == kLim) ? ZW_INC : ZW_CONT;
            break;
        }
        ZWPH(out, WPH_SEARCH);
        // the repcode probe of an iteration comes before its two matches (:268-290, then :292-299 / :317-326)
        uint32_t const isRep = ((jr - i - 2) >> 1) <= ((jm - i) >> 1) ? 1u : 0u;
        uint32_t const j = isRep ? jr : jm;
#endif
        uint32_t const c = __builtin_amdgcn_readlane(candSel, (int)j);
        uint32_t const room = B + j - anchor;
        uint32_t const limit = isRep ? 1u : (room < c - prefixLow ? room : c - prefixLow);        // :271 / :387 (match0 > prefixStart)
        uint32_t const off = isRep ? rep1 : B + j - c;
        // lanes i .. j are inserted (hash0 = ip0); a repcode is found at ip2 = j, i.e. up to j-1 = ip1 (:283); a match
        // also inserts ip1 = j+1 (:296, :323 step <= 4)
        INS |= ZHIP_SBFM64(j + 1 - isRep - i, i);
        if (!isRep) {
            INS |= 1ull << (j + 1);
            rep2 = rep1; x2 = x1; rep1 = off;
            x1 = 1;
            if (P >= off) x1 = cur32 ^ ld32(src + (P - off));
        }
        unsigned long long const E1b = ZW_Eb(x1);
        ZWPH(out, WPH_M_ELOAD);
        uint32_t run = 0;
        if (j) {
            unsigned long long const t = ~E1b << (64 - j);
            run = t ? (uint32_t)__clzll((long long)t) : j;
        }
        if (run == j && limit > run) run += wave_count_back(src, B, B - off, limit - run);
        uint32_t const back = run < limit ? run : limit;
        uint32_t const fl = fwd_run(src, nm8, B, E1b, j + 4, off);
        uint32_t const mLength = 4 + back + fl;
        ZWPH(out, WPH_M_RUNS);
        uint32_t const ll = room - back;
        ZW_EMIT(ll, isRep ? 1u : off + 3, mLength);
        sumLit += ll;
        uint32_t sL = j - back;
        if (back > j) { backBefore = back - j; sL = 0; }
        uint32_t e = j + 4 + fl;
        anchor = B + e;
        uint32_t const cur0L = j - 2 * isRep;
        if (e >= 64) {                                                    // ran past the window
            COV |= lanes_from(sL);
            ZW_LEAVE(e, 1u, B + cur0L);
            if (carry_) INS |= 1ull << (cur0L + 2);                       // :407 is a lane of this window (cur0L <= 59); :408 and :410 are the next window's lanes 0 and 2
            ZWPH(out, WPH_LEAVE);
            break;
        }
        COV |= ZHIP_SBFM64(e - sL, sL);
        INS |= (1ull << (cur0L + 2)) | (1ull << (e - 2));                 // :407-408 (ip0 <= ilimit inside a window)
        if (rep2 && ((ZW_Eq(x2) >> e) & 1)) {                             // :410-420
            ZW_IMMEDIATE(e);
            if (e >= 64) { ZW_LEAVE(e, 2u, 0u); break; }
        }
        i = e;
        kLim = 64; nextStep = B + e + 128;                                // _start: a fresh scan inside the window
        ZWPH(out, WPH_M_EMIT);
    }
window_done:
#undef ZW_IMMEDIATE
#undef ZW_LEAVE
#undef ZW_Eq
#undef ZW_Eb
    ZWPH(out, WPH_SEARCH);
#undef ZW_EMIT
#ifdef ZHIP_DBG_PRINT
    if (lane == 0) printf("  window B=%u carry=%u -> i=%u status=%d INS=%llx NF=%llx COV=%llx nbSeq=%u rep=%u/%u anchor=%u carryOut=%u\n", B, carryIn, i, status, INS, NF, COV, out.nbSeq, rep1, rep2, anchor, carry_);
#endif
    if (ZHIP_FAST_PRELOAD) {
        // the next window's source bytes, requested before this window's stores; unconditional (a load inside a branch is waited for inside
        // the branch): without a next window the lanes read their own bytes again
        bool const nxt = status != ZW_INC && B + i + ZHIP_WIN_NEED <= n;
        uint32_t const q = (nxt ? B + i : B) + lane;
        pre.c8 = ld64(src + q);
        pre.v1 = ld32(src + (q - (nxt ? rep1 : 0u))); pre.v2 = ld32(src + (q - (nxt ? rep2 : 0u)));
        pre.B = nxt ? B + i : ~0u;
    }
    ZWPH(out, WPH_E_PRE);
    ZW_TABLE_FLUSH();
    ZWPH(out, WPH_E_TAB);
#undef ZW_TABLE_FLUSH
    {   // its literals: the lanes behind the new scan position that no match covers (the tail after the last match is
        // tentative: a later backward extension may take it back, its bytes are then simply overwritten)
        unsigned long long LIT = i < 64 ? (~COV & lanes_below(i)) : ~COV;
        if (carryIn == 1) LIT &= ~3ull;
        if (__builtin_amdgcn_inverse_ballot_w64(LIT)) {
            uint32_t const before = __builtin_amdgcn_mbcnt_hi((uint32_t)(COV >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)COV, 0));
            out.lits[out.litPos + (B - anchorEntry) + lane - before - backBefore] = (uint8_t)cur8;
        }
    }
    out.litPos += sumLit;
    ip0_ = B + i; anchor_ = anchor; rep1_ = rep1; rep2_ = rep2;
    ZWPH(out, WPH_E_OUT);
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
#ifdef ZHIP_PROF
    uint32_t ws_acc_[32], wh_acc_[32]; uint64_t zp_last_ = __builtin_amdgcn_s_memtime();
    for (int i_ = 0; i_ < 32; i_++) { ws_acc_[i_] = 0; wh_acc_[i_] = 0; }
    out.ws = ws_acc_; out.wh = wh_acc_; out.zlast = &zp_last_;
#endif
    ZWPH(out, WPH_INIT);

    uint32_t anchor = b0, rep1 = repIn1, rep2 = repIn2, saved1 = 0, saved2 = 0;
    // :238-244  a repcode that reaches below the window is set aside for the block
    if (rep2 > maxRep) { saved2 = rep2; rep2 = 0; }
    if (rep1 > maxRep) { saved1 = rep1; rep1 = 0; }

    if (n - b0 >= 12) {                     // shorter blocks never run an iteration (ip3 = ip0 + 2 + 1 < n - 8); keeps n - 8 >= b0
    uint32_t const nm8 = n - 8;
    int32_t const ilimit = (int32_t)nm8;
    uint32_t ip0 = b0 + (b0 == prefixLow);                                   // :238 ip0 += (ip0 == prefixStart)
    if (ip0 != b0 && lane == 0) lits[0] = src[b0];                           // that position is never searched: windows only store their own lanes
    // (the schedule's lane offsets are recomputed where a batch needs them — six instructions — instead of riding through every window in two registers)
    unsigned long long const evenLanes = 0x5555555555555555ull;

    uint32_t carry = 0;                     // a window handed the end of its last match to the next one (see window_batch)
    FastPre pre; pre.c8 = 0; pre.v1 = 0; pre.v2 = 0; pre.B = ~0u;
    bool have = false;                      // `cur` already holds the bytes of the batch that starts at ip0
    FastBatch cur; cur.bytes = 0; cur.rcur = 0; cur.rv = 0;
    for (;;) {                                                               // one turn per `_start`
        uint32_t step = stepSize, g0 = stepSize, nextStep = ip0 + 128;
        if ((int32_t)(ip0 + g0 + 1) >= ilimit) break;                        // :257
        uint32_t posOff = 0, rposOff = 0; bool haveOffs = false;

        // ---- scan until an event or the end of the unit: windows while the gap is 2, schedule-shaped batches otherwise
        int evKind = 0;                      // 0 none (unit finished), 1 match, 2 repcode, 3 the window handled its events
        uint32_t mpos = 0, cand0 = 0, cur0 = 0;
        bool dense = ZHIP_WIN_DENSE_ONLY == 0;   // windows right behind an event; a scan that found nothing in a whole window goes on in
        dense = true;                            // schedule-shaped batches (cheaper per position when events are far apart)
        for (;;) {
            if (dense && stepSize == 2 && step == 2 && g0 == 2 && ip0 + ZHIP_WIN_NEED <= n) {
                ZWPH(out, WPH_LOOP);
                int const st = window_batch<MLS, TAB>(src, n, nm8, hshift, T, out, ip0, anchor, rep1, rep2, nextStep, prefixLow, carry, pre);
                have = false;
                if (st == ZW_RESTART) { evKind = 3; break; }
                if (st == ZW_INC) { step = 3; nextStep += 128; batch_offsets(g0, step, posOff, rposOff); haveOffs = true; }
                if (ZHIP_WIN_DENSE_ONLY) dense = false;
                continue;
            }
            if (!haveOffs) { batch_offsets(g0, step, posOff, rposOff); haveOffs = true; }
            if (!have) cur = batch_load(src, nm8, ip0, posOff, rposOff, rep1);
            have = false; pre.B = ~0u;
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
            ZWPH(out, WPH_B_SCAN);

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
        ZWPH(out, WPH_B_SCAN);
        wave_extend(src, nm8, mpos, cand0, lim, backLen, fwdLen);
        ip0 = mpos - backLen;
        {   uint32_t const mLength = 4 + backLen + fwdLen;
            lits_copy_now(out, src, nm8, anchor, ip0 - anchor);
            store_seq(out, ip0 - anchor, offBase, mLength);
            ip0 += mLength; anchor = ip0;
        }
        ZWPH(out, WPH_B_MATCH);
        // ---- :403-420 complementary inserts + immediate repcode; the next batch's bytes ride along
        have = false;
        if ((int32_t)ip0 <= ilimit)
        {   // the next batch's bytes ride along with the post-match round only when a batch can follow: behind a match the gap is stepSize again, and with stepSize 2 a WINDOW
            // takes over (it loads its own bytes), so `cur` is not kept alive through the windows for nothing
            if (stepSize == 2) have = post_match<MLS, false, TAB>(src, nm8, hshift, T, out, ip0, anchor, rep1, rep2, cur0, true, 0, 0, cur);
            else {
                uint32_t sp, sr; batch_offsets(stepSize, stepSize, sp, sr);
                have = post_match<MLS, true, TAB>(src, nm8, hshift, T, out, ip0, anchor, rep1, rep2, cur0, true, sp, sr, cur);
            }
        }
        ZWPH(out, WPH_B_POST);
    }
    lits_copy_now(out, src, nm8, anchor, n - anchor);                       // trailing literals (zstd_compress.c:3365)
    } else {
        for (uint32_t i = lane; i < n - b0; i += 64) lits[i] = src[b0 + i]; // tiny block: everything is a literal
        out.litPos = n - b0;
    }
    // ---- _cleanup (:368-375)
    ZWPH(out, WPH_TAIL);
#ifdef ZHIP_PROF
    if (threadIdx.x == 0) for (int i_ = 0; i_ < 32; i_++) { atomicAdd(&zhip::g_wph[i_], (unsigned long long)ws_acc_[i_]); atomicAdd(&zhip::g_wph[32 + i_], (unsigned long long)wh_acc_[i_]); }
#endif
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
