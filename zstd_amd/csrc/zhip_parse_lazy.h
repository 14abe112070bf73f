// zhip_parse_lazy.h — gfx950 hash-chain match finder + greedy / lazy / lazy2 parser (strategies 3..5), in three kernels.
//
// WHAT it computes: exactly the sequences the reference's ZSTD_compressBlock_greedy / _lazy / _lazy2
// (lib/compress/zstd_lazy.c:1516-1779 with search_hashChain = ZSTD_HcFindBestMatch :667-773 and
// ZSTD_insertAndFindFirstIndex_internal :632-657) emit for a unit with no history.
//
// HOW (CDNA4 design, not a translation).  The reference interleaves three things per searched position: insert the
// positions passed so far into a hash table + chain table, walk up to 2^searchLog chain candidates, and decide.  Two
// of them do not depend on the parse:
//   * unless the parser is in its "lazy skipping" mode (:1613-1624, incompressible stretches), EVERY position before
//     the searched one has been inserted, so the chain link of position p is simply "the closest earlier position with
//     the same hash": one array prev[p] per unit, built once               -> k_hc_chain  (one wavefront per unit,
//     the hash heads in LDS, one slice of the hash space per sweep);
//   * with all links known, ZSTD_HcFindBestMatch(p) is a pure function of p: best[p] = (length, offset) for every
//     position of the unit, one GPU thread per position, no ordering at all -> k_hc_search (256 positions per workgroup,
//     workgroups of one unit kept on one XCD so that the unit's source + links stay in that XCD's L2);
//   * what remains sequential is the parser proper: repcode probes, the lazy "is the next position better" comparisons,
//     catch-up, the immediate-repcode loop and the sequence store        -> k_parse_lazy (one wavefront per unit): 64
//     lanes look at the next 64 scheduled positions at once (best[] record + repcode probe per lane), a ballot finds
//     the first event in the reference's order, and the lazy look-ahead reads its neighbours' records from registers.
// Exactness in the two cases where best[] is not what the reference would have found:
//   * lazy-skipping leaves positions un-inserted.  The parser flags them in prev[] (top bit) and remembers the highest
//     one; a record whose walk reached down to a flagged region (its lowest candidate <= that mark) is redone live,
//     walking prev[] and stepping over flagged positions without counting them as attempts;
//   * k_hc_search caps its byte compares at ZHIP_HC_CAP (32) bytes per candidate — the compare loops of 64 unrelated
//     positions diverge and every lane pays for the longest one, and a unit of zeros would cost n^2.  A candidate that
//     reaches the cap beats every shorter one, so the record just names the (one or two) capped candidates and the parser
//     measures them with wave-wide compares; three or more (runs) -> the parser redoes the walk live.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_parse.h"
#include "zhip_parse_dfast.h"
#include "zhip_tables.h"

namespace zhip {

#ifndef ZHIP_HC_CAP
#ifdef ZHIP_LZ_STATS        /* emulator builds of tests/tools only: how the searches of the exact parse were served (units: zhip_parse_lazy.h, frames: zhip_frame_lazy.h) */
extern unsigned long long zhip_lz_stats[8];   /* 0 searches, 1 live from the rows, 2 live by walking links, 3 links followed, 4 catch-up steps, 5 of them without an insert */
#define LZ_STAT(i, n) do { if (lane_id() == 0) zhip_lz_stats[i] += (n); } while (0)
#else
#define LZ_STAT(i, n) do { } while (0)
#endif
#define ZHIP_HC_CAP 32u
#endif
#define ZHIP_HC_NONE     0x1FFFFu            /* "no candidate" in the minCand field */
#define ZHIP_HC_SKIPPED  0x80000000u         /* prev[] flag: position was never inserted (lazy skipping) */
#define ZHIP_HC_PRED     0x40000000u         /* prev[] flag, row matcher: a first parse in predict mode expects this position to be skipped (two-pass prediction, see rh_reconcile) */
#define ZHIP_HC_SEARCH_THREADS 256
#define ZHIP_HC_SEARCH_LDS_THREADS 1024

// per-unit table memory in 32-bit words: prev[ZHIP_UNIT_MAX] (the chain links; the hash heads only ever live in LDS), and for the row matcher its
// row lists (rh_chain_unit): 16 words that are only ever read, then one entry per position
#define ZHIP_RH_LIST_OFF (ZHIP_UNIT_MAX + 16u)
__host__ __device__ inline size_t hc_table_words(uint32_t hashLog) { (void)hashLog; return (size_t)ZHIP_RH_LIST_OFF + ZHIP_UNIT_MAX; }

// best[] record (64 bits): A (17) | B << 17 (17) | mode << 34 (2) | lowest candidate examined << 36 (17)
//   mode 0 exact     : A = offset, B = match length (3 = nothing found)
//   mode 1 / 2 capped: one / two candidates matched at least ZHIP_HC_CAP bytes (A, B = their positions, in chain order) — they
//                      beat every shorter candidate, so the parser only has to measure them (wave-wide compare) and keep the
//                      first longest, exactly what the reference's `if (currentMl > ml)` walk does;
//   mode 3 live      : three or more such candidates (runs, highly repetitive data): the parser redoes the walk
__device__ __forceinline__ uint64_t hc_pack(uint32_t a, uint32_t b, uint32_t mode, uint32_t minCand)
{
    return (uint64_t)a | ((uint64_t)b << 17) | ((uint64_t)mode << 34) | ((uint64_t)minCand << 36);
}
__device__ __forceinline__ uint32_t hc_rec_a(uint64_t r) { return (uint32_t)r & 0x1FFFFu; }
__device__ __forceinline__ uint32_t hc_rec_b(uint64_t r) { return (uint32_t)(r >> 17) & 0x1FFFFu; }
__device__ __forceinline__ uint32_t hc_rec_mode(uint64_t r) { return (uint32_t)(r >> 34) & 3u; }
__device__ __forceinline__ uint32_t hc_rec_min(uint64_t r) { return (uint32_t)(r >> 36) & 0x1FFFFu; }

// ------------------------------------------------------------------ kernel A: chain links, one wavefront per unit
// prev[p] = 1 + the closest q < p with hash(q) == hash(p), 0 if none — what chainTable[p & mask] holds after
// ZSTD_insertAndFindFirstIndex_internal inserted p with every earlier position present (zstd_lazy.c:645-653).
//
// The "most recent position per hash" table has 2^hashLog (2^17) entries of 17 bits: too big for LDS, and in HBM every
// lookup / update is a random 64-byte sector (that version ran at the HBM random-access rate: 50 ms per GiB).  So the
// hash space is cut into slices of 2^14 hashes whose table (16-bit entries + a bit plane for bit 16, 34 KB) fits LDS:
//   1. two cheap scans over the unit (coalesced source reads) bucket the positions by slice, in order, into a queue in
//      global memory (the unit's best[] area, which k_hc_search only writes afterwards): scan one counts, scan two
//      places each entry at its slice's running offset (same-slice rank inside a batch from three ballots);
//   2. one pass per slice streams its queue (coalesced) through the LDS table, 64 entries per step; lanes of a step with
//      equal hash are put in order with the slot-as-detector trick of zhip_parse.h.
// (The version that re-scanned the unit once per slice spent most of its instructions on those scans: 25 ms per GiB.)
#define ZHIP_HC_SLICE_LOG 14u
#ifndef ZHIP_RH_SEEN_BITS
#define ZHIP_RH_SEEN_BITS 16u        /* the row matcher's tag filter, bits per row (rh_chain_unit; 16: two rows share a word; 32 costs the link builder a resident wavefront per CU: 49 KB of LDS instead of 33) */
#endif
#define ZHIP_RH_STAGE_BYTES (2048u + 256u)      /* rh_chain_unit: the source bytes of 32 steps */
// rh_chain_unit's LDS for 2^rl rows: 16-bit cursors | tag filters (together the 32-bit counters of its first pass) | the cursors' 17th bits | staged source bytes
__host__ __device__ inline uint32_t rh_chain_lds_need(uint32_t rl)
{
    uint32_t const a = (2u << rl) + (ZHIP_RH_SEEN_BITS / 8u << rl), cnt = 4u << rl, plane = ((1u << rl) >> 3) < 4 ? 4u : ((1u << rl) >> 3);
    return (a > cnt ? a : cnt) + plane + ZHIP_RH_STAGE_BYTES + 64u;
}
__host__ __device__ inline uint32_t hc_chain_lds_bytes(uint32_t hashLog)
{
    uint32_t const e = 1u << (hashLog < ZHIP_HC_SLICE_LOG ? hashLog : ZHIP_HC_SLICE_LOG);
    uint32_t const plane = (e >> 3) < 4 ? 4 : (e >> 3);
    uint32_t const hc = 2u * e + plane + 64u * 4u;        // lo16[e], bit plane, slice counters / cursors
    // the row matcher's link builder (rh_chain_unit) shares the allocation: head table of 2^(hashLog - rowLog) rows (rowLog >= 4) + one 32-bit
    // tag filter per row
    uint32_t const rl = hashLog > 9 ? hashLog - 4 : 5;
    uint32_t const rh = rh_chain_lds_need(rl);
    return hc > rh ? hc : rh;
}

template <uint32_t MLS>
__device__ inline void hc_chain_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, unsigned char* smem,
                                     uint32_t* __restrict__ prev, uint32_t* __restrict__ queue /* >= n entries of scratch */)
{
    if (n < 10) return;                                   // no position is ever searched (ip = 1 < n - 8 fails)
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const nm8 = n - 8, sh = 32 - u.hashLog;
    uint32_t const sliceLog = u.hashLog < ZHIP_HC_SLICE_LOG ? u.hashLog : ZHIP_HC_SLICE_LOG;
    uint32_t const E = 1u << sliceLog, passes = 1u << (u.hashLog - sliceLog);         // passes <= 16: hashLog <= 18 = windowLog + 1 of a 128 KB unit (ZSTD_adjustCParams_internal, zstd_compress.c:1466)
    uint32_t const planeBytes = (E >> 3) < 4 ? 4 : (E >> 3);
    lds_u16* const lo = (lds_u16*)(uintptr_t)smem;
    lds_u32* const hi = (lds_u32*)(uintptr_t)(smem + 2u * E);
    lds_u32* const ctr = (lds_u32*)(uintptr_t)(smem + 2u * E + planeBytes);            // [0..15] counts, [16..31] start offsets, [32..47] cursors (round 6: eight slices until then — hashLog 18 overran them)
    unsigned long long const laneBelow = below_mask((int)lane);

    // ---- 1. bucket the positions 0 .. n-8 by hash slice (the lazy look-ahead searches up to n-8, :1628)
    if (lane < 48) ctr[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base0 = 0; base0 <= nm8; base0 += 512) {                   // scan one: slice sizes
        uint64_t bv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t const p = base0 + 64u * (uint32_t)j + lane, pc = p <= nm8 ? p : nm8;
            bv[j] = MLS <= 4 ? (uint64_t)ld32(src + pc) : ld64(src + pc);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t const p = base0 + 64u * (uint32_t)j + lane;
            if (p <= nm8) __hip_atomic_fetch_add(&ctr[hash_pos<MLS>(bv[j], sh) >> sliceLog], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) { uint32_t acc = 0; for (uint32_t k = 0; k < 16; k++) { ctr[16 + k] = acc; ctr[32 + k] = acc; acc += ctr[k]; } }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t base0 = 0; base0 <= nm8; base0 += 512) {                   // scan two: entries = position | slice index << 17
        uint64_t bv[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t const p = base0 + 64u * (uint32_t)j + lane, pc = p <= nm8 ? p : nm8;
            bv[j] = MLS <= 4 ? (uint64_t)ld32(src + pc) : ld64(src + pc);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t const p = base0 + 64u * (uint32_t)j + lane;
            bool const live = p <= nm8;
            uint32_t const h = hash_pos<MLS>(bv[j], sh), sl = h >> sliceLog;
            // lanes of this batch in the same slice, from four ballots over the slice id's bits
            unsigned long long const b0 = __ballot(sl & 1), b1 = __ballot(sl & 2), b2 = __ballot(sl & 4), b3 = __ballot(sl & 8);
            unsigned long long same = __ballot(live);
            same &= (sl & 1) ? b0 : ~b0; same &= (sl & 2) ? b1 : ~b1; same &= (sl & 4) ? b2 : ~b2; same &= (sl & 8) ? b3 : ~b3;
            uint32_t const rank = (uint32_t)__popcll(same & laneBelow);
            uint32_t const cur = ctr[32 + sl];
            __builtin_amdgcn_wave_barrier();
            if (live) {
                queue[cur + rank] = p | ((h & (E - 1)) << 17);
                if ((same & ~below_mask((int)lane + 1)) == 0) ctr[32 + sl] = cur + rank + 1;     // the slice's last lane moves its cursor
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();

    // ---- 2. one pass per slice: its queue through the LDS table
    for (uint32_t pass = 0; pass < passes; pass++) {
        {   lds_u32* const z = (lds_u32*)(uintptr_t)smem;                       // fresh slice (zstd_compress.c:2020)
            uint32_t const words = (2u * E + planeBytes) >> 2;
            for (uint32_t i = lane; i < words; i += 64) z[i] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t const q0 = ctr[16 + pass], q1 = q0 + ctr[pass];
        for (uint32_t qb = q0; qb < q1; qb += 64) {
            uint32_t const cnt = q1 - qb < 64 ? q1 - qb : 64;
            bool const live = lane < cnt;
            uint32_t const e = live ? queue[qb + lane] : 0;
            uint32_t const p = e & 0x1FFFFu, idx = e >> 17;
            bool const anyHigh = __builtin_amdgcn_readlane(p, (int)cnt - 1) >= 65535u;     // positions ascend inside a slice queue
            uint32_t old = lo[idx];
            if (anyHigh) old |= ((hi[idx >> 5] >> (idx & 31)) & 1u) << 16;
            __builtin_amdgcn_wave_barrier();
            if (live) lo[idx] = (uint16_t)lane;
            __builtin_amdgcn_wave_barrier();
            unsigned long long const liveMask = below_mask((int)cnt);
            unsigned long long const lose = __ballot(live && lo[idx] != (uint16_t)lane);
            uint32_t cand = old;
            unsigned long long grp = 0;
            if (lose) {
                grp = lane_groups(idx, lose, liveMask);
                unsigned long long const before = grp & laneBelow;
                uint32_t const pd = before ? 63u - (uint32_t)__clzll((long long)before) : lane;
                uint32_t const dp = __shfl(p, (int)pd);
                if (before) cand = dp + 1;
            }
            __builtin_amdgcn_wave_barrier();
            if (live) {
                prev[p] = cand;
                if ((grp & ~below_mask((int)lane + 1)) == 0) {                      // the last lane of a hash group wins
                    uint32_t const v = p + 1;
                    lo[idx] = (uint16_t)v;
                    if (v >> 16) __hip_atomic_fetch_or(&hi[idx >> 5], 1u << (idx & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ------------------------------------------------------------------ kernel B: best match per position, one thread each
// ZSTD_HcFindBestMatch (zstd_lazy.c:667-773, noDict) at position p with every earlier position inserted.
__device__ inline uint64_t hc_search_pos(const uint8_t* __restrict__ src, uint32_t n, uint32_t p,
                                         const uint32_t* __restrict__ prev, uint32_t searchLog, uint32_t chainLog)
{
    uint32_t const nm8 = n - 8, chainSize = 1u << chainLog;
    uint32_t attempts = 1u << searchLog;
    uint32_t ml = 3, off = 0, minCand = ZHIP_HC_NONE, nCap = 0, capA = 0, capB = 0;
    uint32_t m = prev[p];
    while (m != 0 && attempts) {
        uint32_t const mp = m - 1;
        uint32_t const nx = prev[mp];                                           // independent of the compare below
        minCand = mp;
        if (nCap < 3 && p + ml < n && ld32(src + mp + ml - 3) == ld32(src + p + ml - 3)) {   // :714 quick reject at the current best length (a longer match needs byte ml); with three capped candidates the record says "live" whatever follows
            uint32_t cur = 0;
            for (;;) {
                uint32_t const same = lane_same_fwd(src, p + cur, p - mp, nm8);
                cur += same;
                if (same < 8 || cur >= ZHIP_HC_CAP) break;
            }
            if (cur >= ZHIP_HC_CAP) { if (nCap == 0) capA = mp; else if (nCap == 1) capB = mp; nCap++; if (ml < ZHIP_HC_CAP) ml = ZHIP_HC_CAP; }
            else if (cur > ml) { ml = cur; off = p - mp; if (p + cur == n) break; }    // :724-728
        }
        if (p >= chainSize && mp <= p - chainSize) break;                       // :732 matchIndex <= minChain
        m = nx; attempts--;
    }
    if (nCap == 0) return hc_pack(off, ml, 0, minCand);
    if (nCap <= 2) return hc_pack(capA, capB, nCap, minCand);
    return hc_pack(0, 0, 3, minCand);
}

// The same search with the unit's source staged in LDS (k_hc_search_lds: one 1024-thread workgroup per unit, the whole
// <= 128 KB window in the CU's 160 KB of LDS).  The three byte compares per chain step then cost LDS reads instead of fully
// divergent global gathers — the texture-addresser work that bounded the first version; only the chain links stay in L2/HBM.
typedef ZHIP_LDS uint32_t __attribute__((aligned(1))) lds_u32_unal;
typedef ZHIP_LDS uint64_t __attribute__((aligned(1))) lds_u64_unal;
__device__ __forceinline__ uint32_t lds_ld32(const lds_u8* p) { return *(const lds_u32_unal*)p; }
__device__ __forceinline__ uint64_t lds_ld64(const lds_u8* p) { return *(const lds_u64_unal*)p; }
// equal leading bytes (0..8) of the 8-byte windows at q and q-off; the LDS copy has 16 zero bytes after the unit's end
__device__ __forceinline__ uint32_t lds_same_fwd(const lds_u8* lsrc, uint32_t q, uint32_t off, uint32_t n)
{
    uint64_t const x = lds_ld64(lsrc + q) ^ lds_ld64(lsrc + (q - off));
    uint32_t const lim = n - q < 8 ? n - q : 8;                            // q < n
    uint32_t const same = x ? (uint32_t)(__ffsll((long long)x) - 1) >> 3 : 8;
    return same < lim ? same : lim;
}
__device__ inline uint64_t hc_search_pos_lds(const lds_u8* lsrc, uint32_t n, uint32_t p, const uint32_t* __restrict__ prev,
                                             uint32_t searchLog, uint32_t chainLog)
{
    uint32_t const chainSize = 1u << chainLog;
    uint32_t attempts = 1u << searchLog;
    uint32_t ml = 3, off = 0, minCand = ZHIP_HC_NONE, nCap = 0, capA = 0, capB = 0;
    uint32_t m = prev[p];
    while (m != 0 && attempts) {
        uint32_t const mp = m - 1;
        uint32_t const nx = prev[mp];
        minCand = mp;
        if (nCap < 3 && p + ml < n && lds_ld32(lsrc + mp + ml - 3) == lds_ld32(lsrc + p + ml - 3)) {      // (three capped candidates: the record says "live" whatever follows — round 6: runs at level 10 spent 70 ms per 64 MiB here)
            uint32_t cur = 0;
            for (;;) {
                uint32_t const same = (p + cur < n) ? lds_same_fwd(lsrc, p + cur, p - mp, n) : 0;
                cur += same;
                if (same < 8 || cur >= ZHIP_HC_CAP) break;
            }
            if (cur >= ZHIP_HC_CAP) { if (nCap == 0) capA = mp; else if (nCap == 1) capB = mp; nCap++; if (ml < ZHIP_HC_CAP) ml = ZHIP_HC_CAP; }
            else if (cur > ml) { ml = cur; off = p - mp; if (p + cur == n) break; }
        }
        if (p >= chainSize && mp <= p - chainSize) break;
        m = nx; attempts--;
    }
    if (nCap == 0) return hc_pack(off, ml, 0, minCand);
    if (nCap <= 2) return hc_pack(capA, capB, nCap, minCand);
    return hc_pack(0, 0, 3, minCand);
}

// ------------------------------------------------------------------ the row-hash matcher (zstd_lazy.c:778-960, :1141-1340)
// the reference's DEFAULT for greedy / lazy / lazy2 when windowLog > 14 (zstd_compress.c:237-253).  A row keeps the
// (2^rowLog - 1) most recently inserted positions whose salted hash has the same upper bits, with the hash's low 8 bits as a
// tag; a search looks at the row's entries with its own tag, most recent first, at most 2^min(searchLog, rowLog) of them.
// With every earlier position inserted that is again a pure function of the position: "the previous position of the same row"
// is a chain link (prev[], built like the hash-chain links but keyed by the row index — 2^(hashLog - rowLog) <= 2^13 heads, so
// the head table is the 17-bit LDS table of zhip_parse.h and needs no slicing), the tag rides in the link word, and the
// search walks at most 2^rowLog - 1 links.  The salt is the one a fresh CCtx has on its first frame.
// (round 5) The links became LISTS: a walk down 15 links is 15 dependent loads from 15 cache lines, and k_hc_search_lds was bound by exactly
// those requests (two walks per thread in flight made it slower, profiles/r05_ab_l5_rowlists.log).  The builder now sorts the unit's positions by
// row — count, prefix sum, then in position order every position takes the next index of its row — into rowList[] (behind prev[], ZHIP_RH_LIST_OFF),
// so the row's earlier positions lie right below a position's own index, most recent first going down: 64 contiguous bytes for a whole row.
//   prev[p]     = index of p in rowList | ZHIP_RH_FIRST (p opens its row) | tag(p) << 18 | ZHIP_RH_TAGSEEN   (| ZHIP_HC_SKIPPED / ZHIP_HC_PRED from the parser)
//   rowList[i]  = position | tag << 17 | ZHIP_RL_FIRST (the row's first position: a walk ends behind it)
#define ZHIP_RH_IDX_MASK 0x1FFFFu
#define ZHIP_RH_FIRST    0x20000u
#define ZHIP_RL_FIRST    0x02000000u
// (round 5) prev[p] bit 26, copied into bit 53 of p's record: some EARLIER position of p's row may carry p's tag (a 32-bit filter per row over the
// tags' low five bits, kept by the link builder over ALL earlier positions of the row).  Clear = no position the row ever saw has p's tag, so
// ZSTD_RowFindBestMatch at p finds no candidate whatever subset of them the parse inserted: such a search never needs to be redone live — on
// long-match data most live searches were of this kind and failed (profiles/r05_l5_phases_before.log: 17 000 live searches per unit, 12 300 failed).
#define ZHIP_RH_TAGSEEN  0x04000000u
#define ZHIP_REC_TAGSEEN (1ull << 53)
#define ZHIP_REC_WHOLE (1ull << 54)      /* row matcher: the record's walk saw every earlier position of its row (see rh_search_pos_lds) */
__host__ __device__ inline uint64_t rh_bitmix(uint64_t val, uint64_t len)            // zstd_compress.c:1964-1970
{
    val ^= ((val >> 49) | (val << 15)) ^ ((val >> 24) | (val << 40));
    val *= 0x9FB21C651E98DF25ULL;
    val ^= (val >> 35) + len;
    val *= 0x9FB21C651E98DF25ULL;
    return val ^ (val >> 28);
}
__host__ __device__ inline uint64_t rh_fresh_salt() { return rh_bitmix(0, 8) ^ rh_bitmix(0, 4); }   // ZSTD_advanceHashSalt from (0, 0), :1973
// ZSTD_hashPtrSalted (zstd_compress_internal.h:820-879), hBits <= 32: the top hBits of (product ^ salt)
template <uint32_t MLS>
__device__ __forceinline__ uint32_t hash_pos_salted(uint64_t bytes, uint32_t hBits, uint64_t salt)
{
    if (MLS <= 4) return (((uint32_t)bytes * 2654435761U) ^ (uint32_t)salt) >> (32 - hBits);
    uint32_t const top = MLS == 5 ? mulhi64_top32(bytes, 889523592379ULL << 24) : mulhi64_top32(bytes, 227718039650203ULL << 16);
    return (top ^ (uint32_t)(salt >> 32)) >> (32 - hBits);
}
typedef ZHIP_LDS uint16_t __attribute__((may_alias)) lds_u16_alias;
typedef ZHIP_LDS uint32_t __attribute__((may_alias)) lds_u32_alias;

template <uint32_t MLS>
__device__ inline void rh_chain_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, unsigned char* smem, uint32_t* __restrict__ prev)
{
    if (n < 10) return;
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const nm8 = n - 8, rowHashLog = (uint32_t)u.hashLog - u.rowLog, hBits = rowHashLog + 8, rows = 1u << rowHashLog;
    uint64_t const salt = rh_fresh_salt();
    uint32_t* const rowList = prev + ZHIP_RH_LIST_OFF;
    // LDS (rh_chain_lds_need): first the rows' 32-bit counts; then, in their place, 16-bit cursors (the next index of each row) and the tag filters; the
    // cursors' 17th bits lie behind both
    uint32_t const body = (2u << rowHashLog) + (ZHIP_RH_SEEN_BITS / 8u << rowHashLog) > (4u << rowHashLog) ? (2u << rowHashLog) + (ZHIP_RH_SEEN_BITS / 8u << rowHashLog) : (4u << rowHashLog);
    lds_u32_alias* const cnt = (lds_u32_alias*)(uintptr_t)smem;
    lds_u16_alias* const curLo = (lds_u16_alias*)(uintptr_t)smem;
    lds_u32_alias* const seenBits = (lds_u32_alias*)(uintptr_t)(smem + (2u << rowHashLog));      // per row: which tags (mod ZHIP_RH_SEEN_BITS) it has seen
    lds_u32_alias* const curHi = (lds_u32_alias*)(uintptr_t)(smem + body);
    uint32_t const hiWords = rows < 32 ? 1u : rows >> 5;
    for (uint32_t i = lane; i < (body >> 2) + hiWords; i += 64) cnt[i] = 0;
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    // ---- 1. how many positions each row gets (order does not matter: four steps' loads in flight together)
    for (uint32_t p0 = 0; p0 <= nm8; p0 += 256) {
        uint64_t b[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) { uint32_t const p = p0 + 64 * k + lane, pc = p <= nm8 ? p : nm8; b[k] = MLS <= 4 ? (uint64_t)ld32(src + pc) : ld64(src + pc); }
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            uint32_t const p = p0 + 64 * k + lane;
            if (p <= nm8) __hip_atomic_fetch_add(&cnt[hash_pos_salted<MLS>(b[k], hBits, salt) >> 8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    // ---- 2. every row's first index: prefix sums, 64 rows per step; the 16-bit cursor of row r overwrites the count of row r / 2 (read already)
    {   uint32_t base = 0;
        for (uint32_t r0 = 0; r0 < rows; r0 += 64) {
            uint32_t const r = r0 + lane, c = r < rows ? cnt[r] : 0u;
            uint32_t incl = c;
#pragma unroll
            for (uint32_t d = 1; d < 64; d <<= 1) { uint32_t const t = __shfl_up(incl, d); if (lane >= d) incl += t; }
            uint32_t const start = base + incl - c;
            base += __shfl(incl, 63);
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            if (r < rows) curLo[r] = (uint16_t)start;
            unsigned long long const hiM = __ballot(r < rows && (start >> 16) != 0);
            if (lane == 0) { curHi[r0 >> 5] = (uint32_t)hiM; if (r0 + 32 < rows) curHi[(r0 >> 5) + 1] = (uint32_t)(hiM >> 32); }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
        }
        for (uint32_t i = lane; i < ((ZHIP_RH_SEEN_BITS / 8u << rowHashLog) >> 2); i += 64) seenBits[i] = 0;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
    // ---- 3. positions in order, 64 per step: each takes its row's next index
    unsigned long long const laneBelow = below_mask((int)lane);
    // The source bytes come through LDS, 2048 positions per fetch: a wavefront's loads and stores retire in order, so a load issued after the
    // row-list stores (scattered 4-byte writes) waits for them — with one load per step the loop ran at the pace of those stores (24 ms per GiB
    // instead of 10, profiles/r05_ab_l5_rowlists.log); now one wait per 32 steps, for loads issued 32 steps earlier.
    lds_u32_alias* const stage = (lds_u32_alias*)(uintptr_t)(smem + body + (hiWords << 2));        // ZHIP_RH_STAGE_BYTES
    auto step = [&](uint32_t p0, uint64_t bytes) {
        uint32_t const p = p0 + lane; bool const live = p <= nm8;
        uint32_t const h = hash_pos_salted<MLS>(bytes, hBits, salt), row = h >> 8, tag = h & 0xFFu;
        uint32_t const cur = (uint32_t)curLo[row] | (((curHi[row >> 5] >> (row & 31)) & 1u) << 16);
        // the row's tags BEFORE this step: bit (tag mod SEEN_BITS) of the row's field
        uint32_t const sIdx = ZHIP_RH_SEEN_BITS == 32u ? row : row >> 1, sSh = ZHIP_RH_SEEN_BITS == 32u ? 0u : 16u * (row & 1u);
        uint32_t const sBit = (ZHIP_RH_SEEN_BITS == 32u ? (tag & 31u) : (tag & 15u)) + sSh;
        uint32_t const seenWord = seenBits[sIdx];
        __builtin_amdgcn_wave_barrier();
        if (live) __hip_atomic_fetch_or(&seenBits[sIdx], 1u << sBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (live) curLo[row] = (uint16_t)lane;                                 // the slot as duplicate detector
        __builtin_amdgcn_wave_barrier();
        unsigned long long const liveMask = __ballot(live);
        unsigned long long const lose = __ballot(live && curLo[row] != (uint16_t)lane);
        unsigned long long grp = 0;
        if (lose) grp = lane_groups(row, lose, liveMask);
        __builtin_amdgcn_wave_barrier();
        if (live) {
            uint32_t const rank = (uint32_t)__popcll(grp & laneBelow), total = grp ? (uint32_t)__popcll(grp) : 1u;
            uint32_t const field = ZHIP_RH_SEEN_BITS == 32u ? seenWord : ((seenWord >> sSh) & 0xFFFFu);
            bool const first = rank == 0 && field == 0;                        // every earlier position of the row left a bit in its field
            // a lane that shares its row with an earlier lane of this step counts as "seen" (their tags are not compared: rare, and only costs a live search)
            bool const seen = ((seenWord >> sBit) & 1u) != 0 || rank != 0;
            uint32_t const idx = cur + rank;
            rowList[idx] = p | (tag << 17) | (first ? ZHIP_RL_FIRST : 0u);
            prev[p] = idx | (first ? ZHIP_RH_FIRST : 0u) | (tag << 18) | (seen ? ZHIP_RH_TAGSEEN : 0u);
            if (rank + 1 == total) {                                           // the last lane of a row group leaves the cursor
                uint32_t const nx = cur + total;
                curLo[row] = (uint16_t)nx;
                if (nx >> 16) __hip_atomic_fetch_or(&curHi[row >> 5], 1u << (row & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    // two loops, so that the staged one holds no load whose wait would fall into every step
    uint32_t c0 = 0;
    if (2048 + 256 <= n) {
        Quad nA, nB; uint32_t nC;
        auto fetch = [&](uint32_t base) {                                     // bytes [base, base + 2048 + 256) lie inside the unit
            nA = ld128(src + base + 32 * lane); nB = ld128(src + base + 32 * lane + 16); nC = ld32(src + base + 2048 + 4 * lane);
        };
        fetch(0);
        for (; c0 + 2048 + 256 <= n; c0 += 2048) {
            __builtin_amdgcn_wave_barrier();
            stage[8 * lane] = nA.x; stage[8 * lane + 1] = nA.y; stage[8 * lane + 2] = nA.z; stage[8 * lane + 3] = nA.w;
            stage[8 * lane + 4] = nB.x; stage[8 * lane + 5] = nB.y; stage[8 * lane + 6] = nB.z; stage[8 * lane + 7] = nB.w;
            stage[512 + lane] = nC;
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            fetch(c0 + 2048 + 2048 + 256 <= n ? c0 + 2048 : c0);              // the next fetch (unconditional: a load inside a branch is waited for there)
            for (uint32_t k = 0; k < 2048; k += 64) {
                uint32_t const o = k + lane, sh = (o & 3u) * 8u;
                uint32_t const w0 = stage[o >> 2], w1 = stage[(o >> 2) + 1], w2 = stage[(o >> 2) + 2];
                uint64_t const w01 = (uint64_t)w0 | ((uint64_t)w1 << 32);
                uint64_t const b8 = sh ? (w01 >> sh) | ((uint64_t)w2 << (64u - sh)) : w01;
                step(c0 + k, MLS <= 4 ? (uint64_t)(uint32_t)b8 : b8);
            }
        }
    }
    for (uint32_t p0 = c0; p0 <= nm8; p0 += 64) {                             // the unit's last positions: one load per step
        uint32_t const p = p0 + lane, pc = p <= nm8 ? p : nm8;
        step(p0, MLS <= 4 ? (uint64_t)ld32(src + pc) : ld64(src + pc));
    }
}

// ZSTD_RowFindBestMatch (zstd_lazy.c:1141-1340, noDict) at p with every earlier position inserted; the unit staged in LDS.  The row's earlier
// positions are the entries below p's own in rowList, 16 per fetch (four loads in flight); `havePred`: the predicting parse has marked positions
// in prev[] (k_hc_search_lds's second run) — they take no slot of the row
__device__ inline uint64_t rh_search_pos_lds(const lds_u8* lsrc, uint32_t n, uint32_t p, const uint32_t* __restrict__ prev,
                                             uint32_t searchLog, uint32_t rowLog, bool havePred)
{
    const uint32_t* const rowList = prev + ZHIP_RH_LIST_OFF;
    uint32_t const capped = searchLog < rowLog ? searchLog : rowLog;
    uint32_t attempts = 1u << capped, room = (1u << rowLog) - 1;             // a row holds 2^rowLog - 1 positions (slot 0 is its head byte)
    uint32_t ml = 3, off = 0, minCand = ZHIP_HC_NONE, nCap = 0, capA = 0, capB = 0;
    uint32_t const w0 = prev[p], myTag = (w0 >> 18) & 0xFFu;
    // nothing to walk: p opens its row, or the row never saw p's tag (no candidate whatever is inserted)
    bool on = !(w0 & ZHIP_RH_FIRST) && (w0 & ZHIP_RH_TAGSEEN);
    bool ended = !on, done = false;                                          // ended: the walk passed the row's first position
    uint32_t j = w0 & ZHIP_RH_IDX_MASK;
    auto visit = [&](uint32_t e) {
        uint32_t const mp = e & 0x1FFFFu;
        minCand = mp;                                                       // the lowest position VISITED (predicted-skipped ones included)
        bool const pred = havePred && (prev[mp] & ZHIP_HC_PRED) != 0;       // the predicting parse skipped it: it takes no slot of the row
        if (!pred) {
            room--;
            if (((e >> 17) & 0xFFu) == myTag) {
                attempts--;
                if (!done && nCap < 3 && p + ml < n && lds_ld32(lsrc + mp + ml - 3) == lds_ld32(lsrc + p + ml - 3)) {      // (three capped candidates: "live" whatever follows; the walk goes on for minCand / WHOLE)
                    uint32_t cur = 0;
                    for (;;) {
                        uint32_t const same = (p + cur < n) ? lds_same_fwd(lsrc, p + cur, p - mp, n) : 0;
                        cur += same;
                        if (same < 8 || cur >= ZHIP_HC_CAP) break;
                    }
                    if (cur >= ZHIP_HC_CAP) { if (nCap == 0) capA = mp; else if (nCap == 1) capB = mp; nCap++; if (ml < ZHIP_HC_CAP) ml = ZHIP_HC_CAP; }
                    else if (cur > ml) { ml = cur; off = p - mp; if (p + cur == n) done = true; }      // :1281 best possible: evaluation stops, the row was read anyway
                }
            }
        }
        if (e & ZHIP_RL_FIRST) { ended = true; on = false; }
        else if (!attempts || !room) on = false;
    };
    while (on) {
        Quad const q0 = ld128((const uint8_t*)(rowList + j) - 16), q1 = ld128((const uint8_t*)(rowList + j) - 32),
                   q2 = ld128((const uint8_t*)(rowList + j) - 48), q3 = ld128((const uint8_t*)(rowList + j) - 64);
        uint32_t const E[16] = { q0.w, q0.z, q0.y, q0.x, q1.w, q1.z, q1.y, q1.x, q2.w, q2.z, q2.y, q2.x, q3.w, q3.z, q3.y, q3.x };
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) if (on) visit(E[i]);
        j -= 16;
    }
    // WHOLE: the walk passed the row's first position with room left — the row never held 2^rowLog - 1 positions, so whatever is left out of
    // it the candidates of this search are the inserted positions of its own tag: only a left-out position of the same row AND tag changes it
    uint64_t const seen = ((w0 & ZHIP_RH_TAGSEEN) ? ZHIP_REC_TAGSEEN : 0ull) | ((ended && room) ? ZHIP_REC_WHOLE : 0ull);
    if (nCap == 0) return hc_pack(off, ml, 0, minCand) | seen;
    if (nCap <= 2) return hc_pack(capA, capB, nCap, minCand) | seen;
    return hc_pack(0, 0, 3, minCand) | seen;
}

// ------------------------------------------------------------------ kernel C: the parser, one wavefront per unit
struct HcState {
    uint32_t ntu;           // ms->nextToUpdate (zstd_compress_internal.h:232)
    uint32_t skipping;      // ms->lazySkipping (:253)
    uint32_t gapEnd;        // highest position flagged ZHIP_HC_SKIPPED so far, 0 = none (position 0 is always inserted)
    uint32_t gapFlagged;    // row matcher: where the 384-position rule's flagging of the gap behind nextToUpdate has got to (rh_gap_rule)
    lds_u32* dirty;         // row matcher: one bit per row, set when a position of that row was decided otherwise than predicted (2^(hashLog - rowLog) bits of LDS)
    uint32_t predict;       // 1: the PREDICTING parse — positions it would skip get ZHIP_HC_PRED, nothing is flagged, nothing is stored
    uint32_t scanned;       // exact parse: every position below this has been compared with its prediction
    uint32_t havePred;      // exact parse: a predicting parse ran before it (otherwise nothing is marked and there is nothing to compare)
    uint32_t nLive;         // searches redone live (statistics: ZhipParse.pad0)
    uint32_t budget;        // TRY parse: give up (the unit is parsed again with the prediction) once this many searches went live; 0 = never
    uint32_t abort;
    uint32_t epoch;         // grows whenever a row is marked dirty (what a batch of records looked up about its staleness is then out of date)
};
#define ZHIP_PARSE_REDO 0x5245444Fu          /* ZhipParse.status of a unit whose TRY parse gave up */
#define ZHIP_RH_ROWBITS_BYTES 2048u    /* rows <= 2^14 (hashLog <= 18, rowLog >= 4) */
#ifndef ZHIP_RH_FINE_LOG
#define ZHIP_RH_FINE_LOG 16u           /* then one bit per (row, leading tag bits): the top 16 bits of the row-and-tag hash.  15 bits (6 KB of LDS per wavefront) and five wavefronts
                                          per SIMD instead of four: datagen +1 %, text -3 % (profiles/r05_ab_final.log) — after the row lists the parse no longer follows occupancy */
#endif
#define ZHIP_RH_DIRTY_BYTES (ZHIP_RH_ROWBITS_BYTES + (1u << ZHIP_RH_FINE_LOG) / 8u)
// bit index of a row-and-tag hash of hBits bits in the fine map
__device__ __forceinline__ uint32_t rh_fine_key(uint32_t h, uint32_t hBits) { return hBits > ZHIP_RH_FINE_LOG ? h >> (hBits - ZHIP_RH_FINE_LOG) : h; }

// Two-pass prediction (row matcher).  A record is computed before the parse knows which positions it will leave un-inserted; on
// long-match data the 384-position rule skips the inside of every long match, nearly every row then holds such a position and every
// search would have to be redone live.  So the parse runs twice: first in PREDICT mode — rule logged (ZHIP_HC_PRED), not applied, no live
// searches, nothing stored —, then k_hc_search_lds again, stepping over the predicted positions, then the exact parse, which only
// distrusts a record when a position of its row at or above the record's lowest visited one was DECIDED OTHERWISE than predicted
// (flagged but not predicted: found when it is flagged; predicted but inserted: found by rh_reconcile as the parse passes it).
// Without the first pass every flagged position is such a mismatch: the one-pass behaviour.
template <uint32_t MLS>
__device__ inline void rh_reconcile_t(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, const uint32_t* prev, HcState& st, uint32_t upTo)
{
    uint32_t const nm8 = n - 8, hBits = (uint32_t)u.hashLog - u.rowLog + 8;
    uint64_t const salt = rh_fresh_salt();
    bool any = false;
    for (uint32_t q0 = st.scanned; q0 < upTo; q0 += 64) {
        uint32_t const q = q0 + (uint32_t)lane_id();
        uint32_t const w = q < upTo ? prev[q] : 0;
        bool const mism = (w & ZHIP_HC_PRED) && !(w & ZHIP_HC_SKIPPED);       // predicted skipped, but it was inserted
        if (mism) {
            uint32_t const qc = q < nm8 ? q : nm8;
            uint64_t const bytes = MLS <= 4 ? (uint64_t)ld32(src + qc) : ld64(src + qc);
            uint32_t const row = hash_pos_salted<MLS>(bytes, hBits, salt) >> 8;
            __hip_atomic_fetch_or(&st.dirty[row >> 5], 1u << (row & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        if (__ballot(mism)) any = true;
    }
    if (any) { __builtin_amdgcn_wave_barrier(); if (upTo - 1 > st.gapEnd) st.gapEnd = upTo - 1; st.epoch++; }
    st.scanned = upTo;
}
__device__ inline void rh_reconcile(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, const uint32_t* prev, HcState& st, uint32_t upTo)
{
    if (st.predict || !st.havePred || upTo <= st.scanned) return;
    uint32_t const mls = u.minMatch < 4 ? 4 : (u.minMatch > 6 ? 6 : u.minMatch);
    if (mls == 4) rh_reconcile_t<4>(src, n, u, prev, st, upTo);
    else if (mls == 5) rh_reconcile_t<5>(src, n, u, prev, st, upTo);
    else rh_reconcile_t<6>(src, n, u, prev, st, upTo);
}
// row matcher: the never-inserted positions [f0, f1) — flagged; the rows of those that were not predicted are marked: only searches in
// such a row can differ from their record
template <uint32_t MLS>
__device__ inline void rh_flag_range_t(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, uint32_t* prev, HcState& st, uint32_t f0, uint32_t f1)
{
    uint32_t const nm8 = n - 8, hBits = (uint32_t)u.hashLog - u.rowLog + 8;
    uint64_t const salt = rh_fresh_salt();
    if (st.predict) {
        for (uint32_t q = f0 + (uint32_t)lane_id(); q < f1; q += 64) prev[q] |= ZHIP_HC_PRED;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        return;
    }
    bool any = false;
    for (uint32_t q0 = f0; q0 < f1; q0 += 64) {
        uint32_t const q = q0 + (uint32_t)lane_id();
        bool mism = false;
        if (q < f1) {
            uint32_t const w = prev[q];
            prev[q] = w | ZHIP_HC_SKIPPED;
            mism = !(w & ZHIP_HC_PRED);
            if (mism) {
                uint32_t const qc = q < nm8 ? q : nm8;
                uint64_t const bytes = MLS <= 4 ? (uint64_t)ld32(src + qc) : ld64(src + qc);
                uint32_t const h = hash_pos_salted<MLS>(bytes, hBits, salt), row = h >> 8, fk = rh_fine_key(h, hBits);
                __hip_atomic_fetch_or(&st.dirty[row >> 5], 1u << (row & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_fetch_or(&st.dirty[ZHIP_RH_ROWBITS_BYTES / 4 + (fk >> 5)], 1u << (fk & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        if (__ballot(mism)) any = true;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (any && f1 - 1 > st.gapEnd) st.gapEnd = f1 - 1;
    if (any) st.epoch++;
    if (f1 > st.scanned) st.scanned = f1;
}
__device__ inline void rh_flag_range(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, uint32_t* prev, HcState& st, uint32_t f0, uint32_t f1)
{
    if (f1 <= f0) return;
    rh_reconcile(src, n, u, prev, st, f0);                 // what lies between the last decided position and this range was inserted
    uint32_t const mls = u.minMatch < 4 ? 4 : (u.minMatch > 6 ? 6 : u.minMatch);
    if (mls == 4) rh_flag_range_t<4>(src, n, u, prev, st, f0, f1);
    else if (mls == 5) rh_flag_range_t<5>(src, n, u, prev, st, f0, f1);
    else rh_flag_range_t<6>(src, n, u, prev, st, f0, f1);
}
// is the row of position x dirty?  (uniform x: every lane computes the same)
// `rec`: the record of x — one whose walk saw the whole row (ZHIP_REC_WHOLE) is only out of date when a position of its row AND tag was left out; that
// holds while every left-out position was expected in (no predicting parse: with one, a predicted position that IS inserted adds to the rows)
__device__ inline bool rh_row_dirty(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, const HcState& st, uint32_t x, uint64_t rec)
{
    uint32_t const nm8 = n - 8, hBits = (uint32_t)u.hashLog - u.rowLog + 8, xc = x < nm8 ? x : nm8;
    uint32_t const mls = u.minMatch < 4 ? 4 : (u.minMatch > 6 ? 6 : u.minMatch);
    uint64_t const salt = rh_fresh_salt();
    uint64_t const bytes = mls == 4 ? (uint64_t)ld32(src + xc) : ld64(src + xc);
    uint32_t const h = mls == 4 ? hash_pos_salted<4>(bytes, hBits, salt) : (mls == 5 ? hash_pos_salted<5>(bytes, hBits, salt) : hash_pos_salted<6>(bytes, hBits, salt));
    uint32_t const row = h >> 8;
    if ((rec & ZHIP_REC_WHOLE) && !st.havePred) {
        uint32_t const fk = rh_fine_key(h, hBits);
        return (st.dirty[ZHIP_RH_ROWBITS_BYTES / 4 + (fk >> 5)] >> (fk & 31)) & 1u;
    }
    return (st.dirty[row >> 5] >> (row & 31)) & 1u;
}

// the search the reference would run at x with positions flagged in prev[] missing from the chains (all values uniform)
__device__ inline void hc_search_live(const uint8_t* __restrict__ src, uint32_t n, uint32_t x, const uint32_t* prev,
                                      uint32_t searchLog, uint32_t chainLog, uint32_t& mlOut, uint32_t& offOut)
{
    uint32_t const nm8 = n - 8, chainSize = 1u << chainLog;
    uint32_t attempts = 1u << searchLog;
    uint32_t ml = 3, off = 0;
    uint32_t m = uni(prev[x]) & ~ZHIP_HC_SKIPPED;
    while (m != 0) {
        uint32_t const mp = m - 1;
        uint32_t const w = uni(prev[mp]);
        if (w & ZHIP_HC_SKIPPED) { m = w & ~ZHIP_HC_SKIPPED; continue; }         // never inserted: not a candidate
        if (uni(ld32(src + mp + ml - 3)) == uni(ld32(src + x + ml - 3))) {
            uint32_t const cur = wave_count_fwd(src, x, mp, nm8);
            if (cur > ml) { ml = cur; off = x - mp; if (x + cur == n) break; }
        }
        if (--attempts == 0) break;
        if (x >= chainSize && mp <= x - chainSize) break;
        m = w;
    }
    mlOut = ml; offOut = off;
}

// ZSTD_RowFindBestMatch (zstd_lazy.c:1141-1340) at x with the positions flagged in prev[] missing from the rows, from the row LISTS: x's row as the
// reference holds it is the 2^rowLog - 1 most recent positions below x in the row's list that were not left out.  Lane l looks at the l-th entry
// below x's own (one coalesced load), fetches that position's flag word (one gather), ballots rank the inserted ones, and the candidates — own tag,
// within the row's capacity, at most 2^min(searchLog, rowLog) — are compared at once.  No state is kept between searches (the live rows of round 4
// had to be fed every position the parse passed: 65 GB of scattered stores per GiB, profiles/r05_L5_datagen_sq_tcc.txt).  All results uniform.
__device__ inline void rh_live_lists(const uint8_t* __restrict__ src, uint32_t n, uint32_t x, const uint32_t* prev,
                                     uint32_t searchLog, uint32_t rowLog, uint32_t& mlOut, uint32_t& offOut)
{
    const uint32_t* const rowList = prev + ZHIP_RH_LIST_OFF;
    uint32_t const lane = (uint32_t)lane_id(), nm8 = n - 8, capped = searchLog < rowLog ? searchLog : rowLog;
    uint32_t attempts = 1u << capped, room = (1u << rowLog) - 1;
    uint32_t ml = 3, off = 0;
    uint32_t const w0 = uni(prev[x]), myTag = (w0 >> 18) & 0xFFu;
    uint32_t j = w0 & ZHIP_RH_IDX_MASK;                                       // the entries below j are the row's earlier positions, most recent first going down
    bool more = !(w0 & ZHIP_RH_FIRST), done = false;
    unsigned long long const lower = below_mask((int)lane);
    while (more && attempts && room && !done) {
        bool const inList = lane < j;
        uint32_t const e = rowList[inList ? j - 1u - lane : 0u];
        unsigned long long const firstM = __ballot(inList && (e & ZHIP_RL_FIRST) != 0);
        uint32_t const nRow = firstM ? (uint32_t)first_lane(firstM) + 1u : 64u;      // entries of x's row in this window (its first position ends it)
        bool const inRow = lane < nRow && inList;
        uint32_t const mp = e & 0x1FFFFu;
        uint32_t const w = prev[inRow ? mp : x];
        LZ_STAT(3, 1);
        bool const ins = inRow && !(w & ZHIP_HC_SKIPPED);                     // never inserted: it takes no slot of the row
        unsigned long long const insM = __ballot(ins);
        bool cand = ins && (uint32_t)__popcll(insM & lower) < room && ((e >> 17) & 0xFFu) == myTag;
        unsigned long long const tagM = __ballot(cand);
        cand = cand && (uint32_t)__popcll(tagM & lower) < attempts;
        uint32_t cur = 0;
        if (cand) {
            for (;;) {
                uint32_t const sameB = lane_same_fwd(src, x + cur, x - mp, nm8);
                cur += sameB;
                if (sameB < 8 || cur >= ZHIP_HC_CAP) break;
            }
        }
        unsigned long long rest = __ballot(cand);
        while (rest) {
            int const l = first_lane(rest);
            rest &= rest - 1;
            uint32_t len = (uint32_t)__builtin_amdgcn_readlane(cur, l);
            uint32_t const mpl = (uint32_t)__builtin_amdgcn_readlane(mp, l);
            if (len >= ZHIP_HC_CAP) len = uni(wave_count_fwd(src, x, mpl, nm8));
            if (len > ml) { ml = len; off = x - mpl; if (x + len == n) { done = true; break; } }
        }
        uint32_t const nIns = (uint32_t)__popcll(insM), nTag = (uint32_t)__popcll(tagM);
        room -= nIns < room ? nIns : room;
        attempts -= nTag < attempts ? nTag : attempts;
        more = firstM == 0;
        j -= 64;
    }
    mlOut = ml; offOut = off;
}

// row matcher, a search at x that is NOT in lazy-skipping mode (zstd_lazy.c:916-947): of a gap of more than 384 positions since
// nextToUpdate only the first 96 and the last 32 are inserted — the rest is flagged as never inserted
__device__ inline void rh_gap_rule(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, uint32_t* prev, HcState& st, uint32_t x)
{
    if (x > st.ntu && x - st.ntu > 384) {
        // (a batch start is taken for a search at its first position; when the batch's first event is a repcode taken without a search — greedy — nextToUpdate stays where it
        // was and the NEXT batch start meets the same gap, longer: only what lies behind the part already flagged is new.  Without this a unit of short runs — every
        // sequence a repcode, no search ever — flagged its whole past again at every batch: 4.4 s per unit, profiles/r06_l5_runs_of_24.log)
        uint32_t f0 = st.ntu + 96;
        if (st.gapFlagged > f0) f0 = st.gapFlagged;
        if (x - 32 > f0) { rh_flag_range(src, n, u, prev, st, f0, x - 32); st.gapFlagged = x - 32; }
    }
}

// one ZSTD_HcFindBestMatch / ZSTD_RowFindBestMatch call of the reference at x: insertion bookkeeping + the (pre)computed result
__device__ inline void hc_search(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, uint32_t* prev, HcState& st,
                                 uint32_t x, uint64_t rec, uint32_t& ml, uint32_t& offBase)
{
    if (u.rowLog) {
        // row matcher (zstd_lazy.c:916-947, :1199-1209): ntu = ms->nextToUpdate.  Lazy skipping inserts nothing but the searched
        // position; otherwise everything since ntu goes in — except the middle of a gap of more than 384 positions, of which
        // only the first 96 and the last 32 are inserted
        if (!st.skipping) rh_gap_rule(src, n, u, prev, st, x);
        else if (st.ntu < x) rh_flag_range(src, n, u, prev, st, st.ntu, x);
        st.ntu = x + 1;                                   // the searched position is inserted by the search itself (:1251-1255)
        rh_reconcile(src, n, u, prev, st, x);             // everything below x is decided now: was it what the first pass predicted?
    } else {
    if (st.skipping && st.ntu + 1 < x) {                  // :651 only nextToUpdate itself is inserted; the rest never will be
        for (uint32_t q = st.ntu + 1 + (uint32_t)lane_id(); q < x; q += 64) prev[q] |= ZHIP_HC_SKIPPED;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        st.gapEnd = x - 1; st.epoch++;
    }
    st.ntu = x;
    }
    uint32_t off;
    if (!st.predict) LZ_STAT(0, 1);
    uint32_t const minCand = hc_rec_min(rec), mode = hc_rec_mode(rec);
    bool live = mode == 3 || (st.gapEnd != 0 && minCand != ZHIP_HC_NONE && minCand <= st.gapEnd);
    if (live && u.rowLog && !(rec & ZHIP_REC_TAGSEEN)) live = false;                // no position of x's row ever carried x's tag: the record's "nothing" stands
    if (live && mode != 3 && u.rowLog) live = rh_row_dirty(src, n, u, st, x, rec);      // a record only depends on its own row
    if (live) {
        st.nLive++;
        if (st.budget && st.nLive > st.budget) st.abort = 1;
        if (u.rowLog) { if (!st.predict) LZ_STAT(2, 1); rh_live_lists(src, n, x, prev, u.searchLog, u.rowLog, ml, off); }
        else hc_search_live(src, n, x, prev, u.searchLog, u.chainLog, ml, off);
    }
    else if (mode == 0) { ml = hc_rec_b(rec); off = hc_rec_a(rec); }
    else {                                                // one or two candidates ran into the compare cap: measure them
        uint32_t const nm8 = n - 8, cA = hc_rec_a(rec);
        ml = wave_count_fwd(src, x, cA, nm8); off = x - cA;
        if (mode == 2 && x + ml != n) {                   // :726-728 the walk stops at a match that reaches the end of the block
            uint32_t const cB = hc_rec_b(rec);
            uint32_t const l2 = wave_count_fwd(src, x, cB, nm8);
            if (l2 > ml) { ml = l2; off = x - cB; }
        }
    }
    offBase = off + 3;
}

// predict: the first of the row matcher's two parses (see rh_reconcile) — same walk, but the positions the 384-position rule or lazy
// skipping would leave out are only MARKED in prev[] (ZHIP_HC_PRED); no sequences, literals or meta are written
__device__ inline void parse_lazy_unit(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, unsigned char* smem /* ZHIP_RH_DIRTY_BYTES */,
                                       uint32_t* __restrict__ prev, const uint64_t* __restrict__ best,
                                       ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta, bool predict = false, uint32_t tryBudget = 0, bool havePred = false)
{
    uint32_t const lane = (uint32_t)lane_id();
    {   lds_u32* const z = (lds_u32*)(uintptr_t)smem;
        for (uint32_t i = lane; i < ZHIP_RH_DIRTY_BYTES / 4; i += 64) z[i] = 0;
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t const depth = (uint32_t)u.strategy - 3;      // greedy 0, lazy 1, lazy2 2
    FastOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.longPos = 0; out.longType = 0;
    out.litPos = 0; out.pendV = 0; out.pendSh = 0; out.pendOff = 0; out.pendLen = 0;
    ZPROF_DECL
#ifdef ZHIP_PROF
    out.zp = zp_acc_; out.zlast = &zp_last_;
#endif
    ZPROF(0);

    uint32_t anchor = 0, off1 = 1, off2 = 4, saved1 = 0, saved2 = 0;
    // zstd_lazy.c:1552-1559  ip = 1, lowest index 0 -> maxRep = 1
    if (off2 > 1) { saved2 = off2; off2 = 0; }
    if (off1 > 1) { saved1 = off1; off1 = 0; }

    uint32_t const rowBias = u.rowLog ? 1u : 0u;          // the row matcher's nextToUpdate sits one past the last searched position
    uint32_t nLiveOut = 0;
    if (n >= (u.rowLog ? 18u : 10u)) {
    uint32_t const nm8 = n - 8, ilimit = u.rowLog ? n - 16 : n - 8;          // :1527 the row matcher stops ZSTD_ROW_HASH_CACHE_SIZE earlier
    uint32_t ip = 1;
    HcState st; st.ntu = 0; st.skipping = 0; st.gapEnd = 0; st.gapFlagged = 0; st.dirty = (lds_u32*)(uintptr_t)smem; st.predict = predict ? 1u : 0u; st.scanned = 0; st.nLive = 0; st.budget = u.rowLog ? tryBudget : 0u; st.abort = 0; st.havePred = havePred ? 1u : 0u;
    st.epoch = 0;
    // the batch that will start right after the current sequence (known as soon as its end is: catch-up moves the start,
    // not the end), fetched while the sequence is finished; used if the immediate-repcode loop does not move on
    uint32_t pfIp = 0xFFFFFFFFu, pfOff1 = 0, pfCur4 = 0, pfRv = 0; uint64_t pfRec = 0;
    // with it, the first immediate-repcode test behind that sequence (:1763): the sequence's end and the offset it is tested with are known too, so its
    // two loads travel with the catch-up's instead of making a round trip of their own after it
    uint32_t irAt = 0xFFFFFFFFu, irOff = 0, irCur = 0, irRep = 0;
    auto prefetch = [&](uint32_t at, uint32_t o1, uint32_t o2) {
        uint32_t const xj = at + lane, xc = xj < nm8 ? xj : nm8;
        pfRec = xj < ilimit ? best[xj] : 0;
        pfCur4 = ld32(src + xc + 1); pfRv = ld32(src + (xc + 1 - o1));
        pfIp = at; pfOff1 = o1;
        uint32_t const o2c = o2 <= at ? o2 : 0u;
        irCur = ld32(src + at); irRep = ld32(src + (at - o2c));
        irAt = at; irOff = o2;
    };
    while (ip < ilimit && !st.abort) {                                       // :1581
        uint32_t const step = ((ip - anchor) >> 8) + 1;                      // :1614 kSearchStrength = 8
        // ---- the next position where something happens (repcode hit at x+1, or a search that needs a closer look)
        uint32_t x, K = 0, ip0 = ip;
        uint64_t recj = 0; bool repj = false;
        bool repHit; uint64_t rec;
        unsigned long long ev = 0;
        if (step <= 8) {
            if (u.rowLog) rh_gap_rule(src, n, u, prev, st, ip);       // the batch's first search (at ip, not lazy-skipping) meets the gap since nextToUpdate
            // lanes take the positions the reference visits next while nothing is found: ip, ip+step, ... (same step)
            uint32_t const xj = ip + lane * step;
            bool const valid = xj < ilimit && ((xj - anchor) >> 8) + 1 == step;
            uint32_t const xc = xj < nm8 ? xj : nm8;
            uint32_t cur4, rv;
            if (pfIp == ip && pfOff1 == off1 && step == 1) { recj = pfRec; cur4 = pfCur4; rv = pfRv; }   // loaded while the last sequence was finished
            else { recj = valid ? best[xj] : 0; cur4 = ld32(src + xc + 1); rv = ld32(src + (xc + 1 - off1)); }
            repj = valid && off1 > 0 && rv == cur4;                          // :1600 repcode at ip+1
            if (u.rowLog && havePred) {                                       // the batch's positions (and what lies between them) are inserted by its searches
                uint32_t const Kv = (uint32_t)__popcll(__ballot(valid));
                if (Kv) rh_reconcile(src, n, u, prev, st, ip + (Kv - 1) * step);
            }
            uint32_t const minCand = hc_rec_min(recj);
            bool stale = st.gapEnd != 0 && minCand != ZHIP_HC_NONE && minCand <= st.gapEnd;
            if (u.rowLog && st.gapEnd != 0) {
                stale = valid && stale && rh_row_dirty(src, n, u, st, xc, recj);      // per lane: a record only depends on its own row
            }
            // a search whose row never saw its tag finds nothing whatever the rows hold: never live (row matcher; the hash-chain records do not carry the bit)
            bool const tagSeen = !u.rowLog || (recj & ZHIP_REC_TAGSEEN) != 0;
            bool const needLive = valid && tagSeen && (hc_rec_mode(recj) == 3 || stale);
            bool const found = valid && (hc_rec_mode(recj) != 0 || hc_rec_b(recj) >= 4);
            K = (uint32_t)__popcll(__ballot(valid));
            ev = __ballot(repj || needLive || found);
            ZWPROF_SYNC(out, 1);
            ZWPROF_COUNT(out, 10, 1);
            if (!ev) {                                                       // K failed searches (:1613-1624), lazySkipping = 0
                st.ntu = ip + (K - 1) * step + rowBias; st.skipping = 0;
                ip = ip + K * step;
                continue;
            }
        }
        uint32_t const epoch1 = st.epoch;                                    // what the batch found out about its records' staleness holds while this does not move
        uint32_t matchLength = 0, start = 0, offBase = 1;
        bool direct = false, failed = false;
        // the events of the batch in order: a search that fails (no match, no repcode) moves on to the batch's next event without loading
        // the batch again — on long-match data without the prediction nearly every search is such a live search that fails
        for (;;) {
            int e = 0;
            if (step <= 8) {
                e = first_lane(ev);
                x = ip0 + (uint32_t)e * step;
                if (e > 0) { st.ntu = x - step + rowBias; st.skipping = 0; }
                repHit = (__ballot(repj) >> e) & 1;
                rec = readlane64(recj, e);
            } else {
                x = ip;
                rec = best[x];
                rec = readlane64(rec, 0);
                repHit = off1 > 0 && uni(ld32(src + x + 1)) == uni(ld32(src + (x + 1 - off1)));
            }
            matchLength = 0; start = x + 1; offBase = 1; direct = false;
            if (repHit) {                                                    // :1600-1604
                matchLength = 4 + wave_count_fwd(src, x + 5, x + 5 - off1, nm8);
                if (depth == 0) { direct = true; if (start + matchLength < ilimit) prefetch(start + matchLength, off1, off2); }
            }
            ip = x;
            ZWPROF_SYNC(out, 2);
            if (direct) break;
            {   uint32_t ml2, ob2;                                           // :1607-1611
                uint32_t const live0 = st.nLive;
                hc_search(src, n, u, prev, st, x, rec, ml2, ob2);
                if (st.nLive != live0) { ZWPROF_SYNC(out, 3); ZWPROF_COUNT(out, 12, 1); } else ZWPROF_SYNC(out, 4);
                if (ml2 > matchLength) { matchLength = ml2; start = x; offBase = ob2; }
            }
            if (matchLength >= 4) break;
            ZWPROF_COUNT(out, 13, 1);
            failed = true;                                                   // :1613-1625
            if (step <= 8 && st.epoch == epoch1 && !st.abort) {
                ev &= ~below_mask(e + 1);
                if (ev) { failed = false; continue; }
                st.ntu = ip0 + (K - 1) * step + rowBias; st.skipping = 0;       // the batch's remaining searches fail from their records
                ip = ip0 + K * step;
                break;
            }
            uint32_t const stp = ((x - anchor) >> 8) + 1;
            ip = x + stp;
            st.skipping = stp > 8;                                           // kLazySkippingStep = 8
            break;
        }
        if (failed) continue;
        // records / repcode probes of the positions after x, from the batch registers when they are there
        bool const window = step == 1;
        auto rec_at = [&](uint32_t q) -> uint64_t {
            if (window && q - ip0 < K) return readlane64(recj, (int)(q - ip0));
            uint64_t const r = best[q];
            return readlane64(r, 0);
        };
        auto rep_at = [&](uint32_t q) -> bool {           // MEM_read32(q) == MEM_read32(q - off1), off1 > 0
            if (window && q - 1 - ip0 < K) return (__ballot(repj) >> (q - 1 - ip0)) & 1;
            return uni(ld32(src + q)) == uni(ld32(src + (q - off1)));
        };
        if (!direct) {
            if (depth >= 1) {
                while (ip < ilimit) {                                        // :1628-1700
                    ip++;
                    if (off1 > 0 && rep_at(ip)) {
                        uint32_t const mlRep = 4 + wave_count_fwd(src, ip + 4, ip + 4 - off1, nm8);
                        int const gain2 = (int)(mlRep * 3);
                        int const gain1 = (int)(matchLength * 3 - hb32(offBase) + 1);
                        if (gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                    }
                    {   uint32_t ml2, ob2;
                        hc_search(src, n, u, prev, st, ip, rec_at(ip), ml2, ob2);
                        int const gain2 = (int)(ml2 * 4 - hb32(ob2));
                        int const gain1 = (int)(matchLength * 4 - hb32(offBase) + 4);
                        if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = ob2; start = ip; continue; }
                    }
                    if (depth == 2 && ip < ilimit) {                         // :1663-1698
                        ip++;
                        if (off1 > 0 && rep_at(ip)) {
                            uint32_t const mlRep = 4 + wave_count_fwd(src, ip + 4, ip + 4 - off1, nm8);
                            int const gain2 = (int)(mlRep * 4);
                            int const gain1 = (int)(matchLength * 4 - hb32(offBase) + 1);
                            if (gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                        }
                        {   uint32_t ml2, ob2;
                            hc_search(src, n, u, prev, st, ip, rec_at(ip), ml2, ob2);
                            int const gain2 = (int)(ml2 * 4 - hb32(ob2));
                            int const gain1 = (int)(matchLength * 4 - hb32(offBase) + 7);
                            if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = ob2; start = ip; continue; }
                        }
                    }
                    break;
                }
            }
            if (start + matchLength < ilimit) prefetch(start + matchLength, offBase > 3 ? offBase - 3 : off1, offBase > 3 ? off1 : off2);
            if (offBase > 3) {                                               // :1707-1714 catch up
                uint32_t const off = offBase - 3, match = start - off;
                uint32_t const lim = (start - anchor) < match ? (start - anchor) : match;
                uint32_t const back = wave_count_back(src, start, match, lim);
                start -= back; matchLength += back;
                off2 = off1; off1 = off;
            }
        }
        ZWPROF_SYNC(out, 5);
        if (!predict) {
        lits_copy(out, src, nm8, anchor, start - anchor);                    // :1727-1731
        store_seq(out, start - anchor, offBase, matchLength);
        }
        ZWPROF_COUNT(out, 11, 1);
        ZWPROF(out, 6);
        anchor = ip = start + matchLength;
        st.skipping = 0;                                                     // :1732-1738
        while (ip <= ilimit && off2 > 0) {                                   // :1763-1773
            uint32_t cur4, rep4;
            if (irAt == ip && irOff == off2) { cur4 = irCur; rep4 = irRep; }     // requested before the catch-up
            else { cur4 = ld32(src + ip); rep4 = ld32(src + (ip - off2)); }
            irAt = 0xFFFFFFFFu;
            if (uni(cur4) != uni(rep4)) break;
            uint32_t const rl = 4 + wave_count_fwd(src, ip + 4, ip + 4 - off2, nm8);
            {   uint32_t const t = off2; off2 = off1; off1 = t; }
            if (!predict) store_seq(out, 0, 1, rl);
            ip += rl; anchor = ip;
        }
        ZWPROF_SYNC(out, 7);
    }
    ZPROF_FLUSH(0);
    if (predict) return;
    if (st.abort) { if (lane == 0) meta->status = ZHIP_PARSE_REDO; return; }
    lits_copy(out, src, nm8, anchor, n - anchor);                           // trailing literals (zstd_compress.c:3365)
    lits_flush(out);
    nLiveOut = st.nLive;
    } else {
        if (predict) return;
        for (uint32_t i = lane; i < n; i += 64) lits[i] = src[i];
        out.litPos = n;
    }
    // :1777-1783
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;
    if (lane == 0) {
        meta->nbSeq = out.nbSeq; meta->lastLits = n - anchor;
        meta->longPos = out.longPos; meta->longType = out.longType;
        meta->rep[0] = off1 ? off1 : saved1; meta->rep[1] = off2 ? off2 : saved2; meta->rep[2] = 8;
        meta->status = 0; meta->litSize = out.litPos; meta->pad0 = nLiveOut;      // pad0: searches redone live (statistics only)
    }
}

}  // namespace zhip
