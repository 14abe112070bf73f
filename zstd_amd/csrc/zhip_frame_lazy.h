// zhip_frame_lazy.h — multi-block frames and job-pool frames for the strategies greedy / lazy / lazy2 (levels 5-12 of the large
// size classes), with both of the reference's match finders (hash chain, row hash).
//
// WHAT it computes: the frame ZSTD_compress2 emits for one input above 128 KB — ZSTD_compress_frameChunk (lib/compress/
// zstd_compress.c:4527-4623) driving ZSTD_compressBlock_greedy / _lazy / _lazy2 (zstd_lazy.c:1516-1779) block after block with
// ONE match state: the hash chain (ZSTD_insertAndFindFirstIndex_internal :632-657, ZSTD_HcFindBestMatch :667-773) or the rows
// (ZSTD_row_update :916-947, ZSTD_RowFindBestMatch :1141-1340), ms->nextToUpdate with the "limited update after a very long match"
// rule of ZSTD_buildSeqStore (zstd_compress.c:3243-3249), the window (ZSTD_window_enforceMaxDist :4561, ZSTD_getLowestMatchIndex
// zstd_compress_internal.h:1312), repcodes, the literals' Huffman table and — new with these strategies — the three FSE tables,
// which a later block may repeat when ZSTD_fseBitCost says that is cheapest (zstd_compress_sequences.c:103-135, :205-231).
// Block borders: 128 KB, or 92 KB once the frame has saved 3 bytes; ZSTD_lazy2 places them with the fingerprint splitter of
// zstd_preSplit.c:141-181 (ZSTD_optimalBlockSize, zstd_compress.c:4494-4518).  With a job table the same code produces the frame
// of ZSTD_c_nbWorkers >= 1 (zstdmt_compress.c:683-790, :1168-1233; the lazy strategies index the WHOLE prefix of a job:
// ZSTD_loadDictionaryContent, zstd_compress.c:4878-4965).
//
// HOW (the unit design of zhip_parse_lazy.h stretched over a window of any size).  Per workgroup-unit W (a frame, or a job with
// the overlap in front of it), positions count from the start of W's window:
//   k_lz_links   prev[p] = 1 + the closest earlier position with p's key (its hash, or its row), for every position of the
//                window: one 1024-thread workgroup per W.  The 16 wavefronts share the key space (key mod 16): 1024 keys are hashed
//                once per round into LDS, every wavefront collects its own keys in order until it has 64 of them and then makes ONE
//                gather + ONE scatter on the head table (HBM/L2, 4 << keyBits bytes per W); equal keys inside such a batch are
//                ordered with ballots.  The row matcher's 8-bit tags go to a byte array.
//   k_lz_search  best[p] for every position of W's section, one thread each: the search the reference makes at p when every
//                earlier position is in the table — a pure function of p (the window bound of p is max(0, p - 2^windowLog),
//                whatever the blocks); byte compares are capped at ZHIP_HC_CAP like in the unit kernels.
//   k_frame_lazy one 256-thread workgroup per W walks the blocks: wavefront 0 runs the parser over the block (64 scheduled positions
//                at a time from their records; what the records cannot know is redone live: positions the parser left un-inserted
//                — lazy skipping, the 384/192 rule at block starts, the row matcher's 384-position gaps — are flagged in prev[] and
//                in a dirty-key bitmap; a record within ZHIP_HC_CAP bytes of its block's end was measured against the window's
//                end, not the block's), then the four wavefronts encode the block (zhip_entropy.h) against the previous block's
//                Huffman and FSE tables.
//                Row matcher: those live searches read the rows AS THE REFERENCE KEEPS THEM — per row the 2^rowLog - 1 most recently
//                inserted positions + tags (LzRing), brought up to date by the parser, 64 positions per step, only once a position was
//                left out (data without long matches never pays for it).  A live search is then two round trips (row, candidates' bytes)
//                whatever the number of left-out positions; the walk through prev[] steps over every one of them — on long-match data
//                some 250 dependent loads per search in an 8 MiB window (round 3 / 4: 0.004 GB/s on one job-pool frame).
//   k_lz_predict (on by default, zhip_set_prediction(frames = 0) turns it off) the two-pass prediction of zhip_parse_lazy.h (rh_reconcile) for a window: a first parse marks the positions
//                it would leave un-inserted, k_lz_search runs again stepping over them, and the exact parse distrusts a record only where
//                prediction and truth differ.  Its first 32 KB are a probe: a window that leaves (almost) nothing out there is parsed once.
#pragma once
#include "zhip_parse_lazy.h"
#include "zhip_frame.h"

namespace zhip {

#define ZHIP_LZ_LINK_THREADS 1024
#define ZHIP_LZ_NONE 0xFFFFFFFFu
#define ZHIP_LZ_PRED  0x40000000u             /* prev[] flag: the predicting parse expects this position to stay un-inserted (windows below 2^30 positions) */
#define ZHIP_LZ_LINK(w) ((w) & 0x3FFFFFFFu)

// one per workgroup-unit, filled by the host
struct ZhipLzSlot {
    uint64_t posOff;        // index of W's first position in the prev / tags / best arrays
    uint64_t headOff;       // word offset of W's head table (4 << keyBits bytes; afterwards its first words are the dirty-key bitmap)
    uint32_t span;          // positions of the window: prefix + section
    uint32_t linkStart;     // first position that is ever inserted (a prefix longer than 8 << max(hashLog, chainLog) is only indexed at its end)
    uint32_t holeStart, holeEnd;   // positions of the prefix the reference never inserts (its last 8: zstd_compress.c:4920-4964), empty for a frame
    uint64_t ringOff;       // row matcher: byte offset of W's live rows (lz_ring_bytes) in the ring arena
};

struct LzRec { uint32_t a, b, minCand, mode; };

// host: where unit W's arrays live and which positions of its prefix are indexed — the lazy strategies' part of
// ZSTD_loadDictionaryContent (zstd_compress.c:4878-4965) for a job's raw-content prefix: of a prefix longer than
// 8 << max(hashLog, chainLog) only the suffix is indexed, every position up to its end - 8 goes in, the last 8 never do
__host__ __device__ inline uint32_t lz_key_bits(const ZhipUnit& u);
__host__ __device__ inline uint64_t lz_ring_bytes(const ZhipUnit& u);
__host__ inline void lz_fill_slot(ZhipLzSlot& L, const ZhipUnit& u, uint32_t prefixLen, uint64_t& posCursor, uint64_t& headCursor, uint64_t& ringCursor)
{
    uint32_t const span = prefixLen + u.srcLen;
    L.span = span; L.linkStart = 0; L.holeStart = 0; L.holeEnd = 0;
    L.posOff = posCursor; posCursor += ((uint64_t)span + 16 + 15) & ~(uint64_t)15;
    L.headOff = headCursor; headCursor += (uint64_t)1 << lz_key_bits(u);
    L.ringOff = ringCursor; ringCursor += lz_ring_bytes(u);
    if (prefixLen) {
        uint32_t const big = u.hashLog > u.chainLog ? u.hashLog : u.chainLog;
        uint64_t const maxDict = (uint64_t)8 << (big < 28 ? big : 28);
        uint32_t const ip = prefixLen > maxDict ? (uint32_t)(prefixLen - maxDict) : 0u;
        if (prefixLen - ip <= 8) L.linkStart = prefixLen;
        else { L.linkStart = ip; L.holeStart = prefixLen - 8; L.holeEnd = prefixLen; }
    }
}
      // see hc_pack in zhip_parse_lazy.h; positions and offsets need more than 17 bits here

__host__ __device__ inline uint32_t lz_key_bits(const ZhipUnit& u) { return u.rowLog ? (uint32_t)u.hashLog - u.rowLog : (uint32_t)u.hashLog; }
// the live rows of one W: a count of inserts per row (4 B), then 2^rowLog position slots (4 B) and tag slots (1 B) per row
__host__ __device__ inline uint64_t lz_ring_bytes(const ZhipUnit& u)
{
    if (!u.rowLog) return 0;
    uint64_t const rows = (uint64_t)1 << lz_key_bits(u);
    return ((rows * 4 + (rows << u.rowLog) * 5) + 255) & ~(uint64_t)255;
}
__host__ __device__ inline uint32_t lz_mls(const ZhipUnit& u) { return u.minMatch < 4 ? 4u : (u.minMatch > 6 ? 6u : (uint32_t)u.minMatch); }   // zstd_lazy.c:1531

// key of the 8 bytes at a position: the chain's hash, or the row index (tag = the salted hash's low 8 bits)
template <uint32_t MLS>
__device__ __forceinline__ uint32_t lz_key_t(uint64_t bytes, const ZhipUnit& u, uint32_t& tag)
{
    if (u.rowLog) {
        uint32_t const h = hash_pos_salted<MLS>(bytes, (uint32_t)u.hashLog - u.rowLog + 8, rh_fresh_salt());
        tag = h & 0xFFu;
        return h >> 8;
    }
    tag = 0;
    return hash_pos<MLS>(MLS <= 4 ? (uint64_t)(uint32_t)bytes : bytes, 32 - u.hashLog);
}
__device__ inline uint32_t lz_key(uint64_t bytes, const ZhipUnit& u, uint32_t& tag)
{
    uint32_t const mls = lz_mls(u);
    return mls == 4 ? lz_key_t<4>(bytes, u, tag) : (mls == 5 ? lz_key_t<5>(bytes, u, tag) : lz_key_t<6>(bytes, u, tag));
}

// ------------------------------------------------------------------ k_lz_links
struct LzLinkShared {
    uint32_t key[ZHIP_LZ_LINK_THREADS];                   // keys of the round's positions (ZHIP_LZ_NONE: not a position to insert)
    uint32_t cpos[ZHIP_LZ_LINK_THREADS / 64][128];        // per wavefront: its collected positions ...
    uint32_t ckey[ZHIP_LZ_LINK_THREADS / 64][128];        // ... and their keys, in position order
};

// the first c (<= 64) collected entries of wavefront wv through the head table
__device__ inline void lz_link_batch(LzLinkShared* sh, uint32_t wv, uint32_t c, uint32_t keyBits, uint32_t* __restrict__ prev, uint32_t* __restrict__ heads)
{
    uint32_t const lane = (uint32_t)lane_id();
    bool const live = lane < c;
    uint32_t const p = live ? sh->cpos[wv][lane] : 0u, k = live ? sh->ckey[wv][lane] : 0u;
    uint32_t const old = live ? heads[k] : 0u;
    unsigned long long const liveMask = below_mask((int)c);
    unsigned long long const same = wave_hash_group(k, keyBits) & liveMask;
    unsigned long long const before = same & below_mask((int)lane);
    uint32_t const pd = before ? 63u - (uint32_t)__clzll((long long)before) : lane;
    uint32_t const dp = __shfl(p, (int)pd);
    if (live) {
        prev[p] = before ? dp + 1u : old;
        if ((same & ~below_mask((int)lane + 1)) == 0) heads[k] = p + 1u;              // the last position of a key leaves the head
    }
    __builtin_amdgcn_wave_barrier();
}

template <uint32_t MLS>
__device__ inline void lz_links_t(const uint8_t* __restrict__ src, const ZhipUnit& u, const ZhipLzSlot& L, LzLinkShared* sh,
                                  uint32_t* __restrict__ prev, uint8_t* __restrict__ tags, uint32_t* __restrict__ heads)
{
    uint32_t const t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    uint32_t const keyBits = lz_key_bits(u);
    for (uint32_t i = t; i < (1u << keyBits); i += ZHIP_LZ_LINK_THREADS) heads[i] = 0;         // fresh table (zstd_compress.c:2020)
    __syncthreads();
    if (L.span < 9 || L.linkStart > L.span - 8) return;
    uint32_t const last = L.span - 8;                                                       // the hash reads 8 bytes
    uint32_t cnt = 0;
    for (uint32_t c0 = L.linkStart; c0 <= last; c0 += ZHIP_LZ_LINK_THREADS) {
        uint32_t const p = c0 + t;
        uint32_t k = ZHIP_LZ_NONE;
        if (p <= last && !(p >= L.holeStart && p < L.holeEnd)) {
            uint32_t tag;
            k = lz_key_t<MLS>(ld64(src + p), u, tag);
            if (u.rowLog) tags[p] = (uint8_t)tag;
        }
        sh->key[t] = k;
        __syncthreads();
        for (uint32_t b = 0; b < ZHIP_LZ_LINK_THREADS / 64; b++) {
            uint32_t const kk = sh->key[b * 64 + lane];
            bool const mine = kk != ZHIP_LZ_NONE && (kk & 15u) == wv;
            unsigned long long const m = __ballot(mine);
            if (m) {
                uint32_t const rank = (uint32_t)__popcll(m & below_mask((int)lane));
                if (mine) { sh->cpos[wv][cnt + rank] = c0 + b * 64 + lane; sh->ckey[wv][cnt + rank] = kk; }
                cnt += (uint32_t)__popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (cnt >= 64) {
                    lz_link_batch(sh, wv, 64, keyBits, prev, heads);
                    uint32_t const rest = cnt - 64;
                    uint32_t const mp = lane < rest ? sh->cpos[wv][64 + lane] : 0u, mk = lane < rest ? sh->ckey[wv][64 + lane] : 0u;
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rest) { sh->cpos[wv][lane] = mp; sh->ckey[wv][lane] = mk; }
                    __builtin_amdgcn_wave_barrier();
                    cnt = rest;
                }
            }
        }
        __syncthreads();
    }
    if (cnt) lz_link_batch(sh, wv, cnt, keyBits, prev, heads);
}

// ------------------------------------------------------------------ k_lz_search: one thread per position
// ZSTD_HcFindBestMatch (zstd_lazy.c:667-773) at p with every earlier position inserted; `end` = the end of W's window
__device__ inline LzRec lz_search_hc(const uint8_t* __restrict__ src, uint32_t end, uint32_t p, const uint32_t* __restrict__ prev,
                                     uint32_t searchLog, uint32_t chainLog, uint32_t lowLimit)
{
    uint32_t const nm8 = end - 8, chainSize = 1u << chainLog;
    uint32_t attempts = 1u << searchLog;
    uint32_t ml = 3, off = 0, minCand = ZHIP_LZ_NONE, nCap = 0, capA = 0, capB = 0;
    uint32_t m = ZHIP_LZ_LINK(prev[p]);
    while (m != 0 && attempts) {
        uint32_t const mp = m - 1;
        if (mp < lowLimit) break;                                                  // :711 matchIndex >= lowLimit
        uint32_t const wl = prev[mp], nx = ZHIP_LZ_LINK(wl);
        minCand = mp;                                                              // the lowest position VISITED
        if (wl & ZHIP_LZ_PRED) { m = nx; continue; }                               // the predicting parse left it out: not in the chain
        if (nCap < 3 && p + ml < end && ld32(src + mp + ml - 3) == ld32(src + p + ml - 3)) {      // (three capped candidates: the record says "live" whatever follows)
            uint32_t cur = 0;
            for (;;) {
                uint32_t const same = lane_same_fwd(src, p + cur, p - mp, nm8);
                cur += same;
                if (same < 8 || cur >= ZHIP_HC_CAP) break;
            }
            if (cur >= ZHIP_HC_CAP) { if (nCap == 0) capA = mp; else if (nCap == 1) capB = mp; nCap++; if (ml < ZHIP_HC_CAP) ml = ZHIP_HC_CAP; }
            else if (cur > ml) { ml = cur; off = p - mp; if (p + cur == end) break; }
        }
        if (p >= chainSize && mp <= p - chainSize) break;                          // :732 matchIndex <= minChain
        m = nx; attempts--;
    }
    LzRec r; r.minCand = minCand;
    if (nCap == 0) { r.a = off; r.b = ml; r.mode = 0; }
    else if (nCap <= 2) { r.a = capA; r.b = capB; r.mode = nCap; }
    else { r.a = 0; r.b = 0; r.mode = 3; }
    return r;
}
// ZSTD_RowFindBestMatch (zstd_lazy.c:1141-1340): the row's entries most recent first, at most 2^rowLog - 1 of them, those with p's tag
__device__ inline LzRec lz_search_rh(const uint8_t* __restrict__ src, uint32_t end, uint32_t p, const uint32_t* __restrict__ prev, const uint8_t* __restrict__ tags,
                                     uint32_t searchLog, uint32_t rowLog, uint32_t lowLimit)
{
    uint32_t const nm8 = end - 8, capped = searchLog < rowLog ? searchLog : rowLog;
    uint32_t attempts = 1u << capped, room = (1u << rowLog) - 1;
    uint32_t ml = 3, off = 0, minCand = ZHIP_LZ_NONE, nCap = 0, capA = 0, capB = 0;
    uint32_t const myTag = tags[p];
    uint32_t m = ZHIP_LZ_LINK(prev[p]);
    bool done = false;
    while (m != 0 && attempts && room) {
        uint32_t const mp = m - 1;
        if (mp < lowLimit) break;                                                  // :1235 (every older entry is below it too)
        uint32_t const wl = prev[mp], w = ZHIP_LZ_LINK(wl); uint32_t const tg = tags[mp];
        minCand = mp;                                                              // the lowest position VISITED
        if (wl & ZHIP_LZ_PRED) { m = w; continue; }                                // the predicting parse left it out: it takes no slot of the row
        room--;
        if (tg == myTag) {
            attempts--;
            if (!done && nCap < 3 && p + ml < end && ld32(src + mp + ml - 3) == ld32(src + p + ml - 3)) {
                uint32_t cur = 0;
                for (;;) {
                    uint32_t const same = lane_same_fwd(src, p + cur, p - mp, nm8);
                    cur += same;
                    if (same < 8 || cur >= ZHIP_HC_CAP) break;
                }
                if (cur >= ZHIP_HC_CAP) { if (nCap == 0) capA = mp; else if (nCap == 1) capB = mp; nCap++; if (ml < ZHIP_HC_CAP) ml = ZHIP_HC_CAP; }
                else if (cur > ml) { ml = cur; off = p - mp; if (p + cur == end) done = true; }
            }
        }
        m = w;
    }
    LzRec r; r.minCand = minCand;
    if (nCap == 0) { r.a = off; r.b = ml; r.mode = 0; }
    else if (nCap <= 2) { r.a = capA; r.b = capB; r.mode = nCap; }
    else { r.a = 0; r.b = 0; r.mode = 3; }
    return r;
}

// ------------------------------------------------------------------ the parser of one block
struct LzRing { uint32_t* cnt; uint32_t* pos; uint8_t* tag; };      // cnt[row] = inserts so far; insert i of a row sits in slot i mod (2^rowLog - 1) of pos / tag[row << rowLog | slot]
struct LzState {
    uint32_t ntu;           // ms->nextToUpdate (hash chain: the last searched position; rows: one past it)
    uint32_t skipping;      // ms->lazySkipping
    uint32_t gapEnd;        // highest position decided otherwise than predicted, 0 = none
    uint32_t* dirty;        // one bit per key: a position with that key was decided otherwise than predicted
    uint32_t predict;       // 1: the PREDICTING parse (zhip_parse_lazy.h: rh_reconcile) — positions it would leave out get ZHIP_LZ_PRED, nothing is flagged or stored
    uint32_t scanned;       // exact parse: every position below this has been compared with its prediction
    uint32_t havePred;      // exact parse: k_lz_predict ran before it (otherwise nothing is marked and there is nothing to compare)
    uint32_t nPred;         // predicting parse: positions marked so far
    LzRing ring;            // row matcher, exact parse: the live rows (cnt == nullptr: none — live searches walk prev[])
    uint32_t ins;           // every position below this that was inserted is in the live rows
    uint32_t nFlagged;      // exact parse: positions left out so far (0: prev[] is the truth and nothing needs the live rows)
    uint32_t epoch;         // exact parse: grows whenever a key is marked dirty (what a batch of records looked up about its staleness is then out of date)
    uint32_t holeStart, holeEnd;   // a job's never-inserted prefix positions (ZhipLzSlot)
    uint32_t gapFlagged;    // row matcher: where the 384-position rule's flagging of the gap behind nextToUpdate has got to (lz_gap_rule)
};
__device__ __forceinline__ LzRec lz_rec_lane(const LzRec& r, int l)
{
    LzRec o;
    o.a = __builtin_amdgcn_readlane(r.a, l); o.b = __builtin_amdgcn_readlane(r.b, l);
    o.minCand = __builtin_amdgcn_readlane(r.minCand, l); o.mode = __builtin_amdgcn_readlane(r.mode, l);
    return o;
}
// exact parse: the positions [st.scanned, upTo) were inserted — those the predicting parse expected to be left out are mismatches
__device__ inline void lz_reconcile(const uint8_t* __restrict__ src, const ZhipUnit& u, const uint32_t* prev, LzState& st, uint32_t upTo)
{
    if (st.predict || !st.havePred || upTo <= st.scanned) return;
    bool any = false;
    for (uint32_t q0 = st.scanned; q0 < upTo; q0 += 64) {
        uint32_t const q = q0 + (uint32_t)lane_id();
        uint32_t const w = q < upTo ? prev[q] : 0;
        bool const mism = (w & ZHIP_LZ_PRED) && !(w & ZHIP_HC_SKIPPED);
        if (mism) { uint32_t tag; uint32_t const k = lz_key(ld64(src + q), u, tag); atomicOr(&st.dirty[k >> 5], 1u << (k & 31)); }
        if (__ballot(mism)) any = true;
    }
    if (any) { __threadfence_block(); __builtin_amdgcn_wave_barrier(); if (upTo - 1 > st.gapEnd) st.gapEnd = upTo - 1; st.epoch++; }
    st.scanned = upTo;
}
// the never-inserted positions [f0, f1) (all below the window's end - 8): flagged; the keys of those that were not predicted are marked
__device__ inline void lz_flag_range(const uint8_t* __restrict__ src, const ZhipUnit& u, uint32_t* prev, LzState& st, uint32_t f0, uint32_t f1)
{
    if (f1 <= f0) return;
    if (st.predict) {
        for (uint32_t q = f0 + (uint32_t)lane_id(); q < f1; q += 64) prev[q] |= ZHIP_LZ_PRED;
        st.nPred += f1 - f0;
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        return;
    }
    lz_reconcile(src, u, prev, st, f0);
    st.nFlagged += f1 - f0;
    bool any = false;
    for (uint32_t q0 = f0; q0 < f1; q0 += 64) {
        uint32_t const q = q0 + (uint32_t)lane_id();
        bool mism = false;
        if (q < f1) {
            uint32_t const w = prev[q];
            prev[q] = w | ZHIP_HC_SKIPPED;
            mism = !(w & ZHIP_LZ_PRED);
            if (mism) { uint32_t tag; uint32_t const k = lz_key(ld64(src + q), u, tag); atomicOr(&st.dirty[k >> 5], 1u << (k & 31)); }
        }
        if (__ballot(mism)) any = true;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    if (any && f1 - 1 > st.gapEnd) st.gapEnd = f1 - 1;
    if (any) st.epoch++;
    if (f1 > st.scanned) st.scanned = f1;
}
__device__ __forceinline__ bool lz_dirty(const uint8_t* __restrict__ src, const ZhipUnit& u, const LzState& st, uint32_t x)
{
    uint32_t tag;
    uint32_t const k = lz_key(ld64(src + x), u, tag);
    return (st.dirty[k >> 5] >> (k & 31)) & 1u;
}
// the hash-chain search the reference makes at x with the flagged positions missing from the chains (all values uniform)
__device__ inline void lz_live_hc(const uint8_t* __restrict__ src, uint32_t bEnd, uint32_t x, const uint32_t* prev, uint32_t searchLog, uint32_t chainLog,
                                  uint32_t lowLimit, uint32_t& mlOut, uint32_t& offOut)
{
    uint32_t const nm8 = bEnd - 8, chainSize = 1u << chainLog;
    uint32_t attempts = 1u << searchLog;
    uint32_t ml = 3, off = 0;
    uint32_t m = ZHIP_LZ_LINK(uni(prev[x]));
    while (m != 0) {
        uint32_t const mp = m - 1;
        if (mp < lowLimit) break;
        uint32_t const w = uni(prev[mp]);
        if (w & ZHIP_HC_SKIPPED) { m = ZHIP_LZ_LINK(w); continue; }
        if (uni(ld32(src + mp + ml - 3)) == uni(ld32(src + x + ml - 3))) {
            uint32_t const cur = (wave_count_fwd(src, x, mp, nm8));
            if (cur > ml) { ml = cur; off = x - mp; if (x + cur == bEnd) break; }
        }
        if (--attempts == 0) break;
        if (x >= chainSize && mp <= x - chainSize) break;
        m = ZHIP_LZ_LINK(w);
    }
    mlOut = ml; offOut = off;
}
__device__ inline void lz_live_rh(const uint8_t* __restrict__ src, uint32_t bEnd, uint32_t x, const uint32_t* prev, const uint8_t* tags, uint32_t searchLog, uint32_t rowLog,
                                  uint32_t lowLimit, uint32_t& mlOut, uint32_t& offOut)
{
    uint32_t const nm8 = bEnd - 8, capped = searchLog < rowLog ? searchLog : rowLog;
    uint32_t attempts = 1u << capped, room = (1u << rowLog) - 1;
    uint32_t ml = 3, off = 0;
    uint32_t const myTag = uni((uint32_t)tags[x]);
    uint32_t m = ZHIP_LZ_LINK(uni(prev[x]));
    bool done = false;
    while (m != 0 && attempts && room) {
        uint32_t const mp = m - 1;
        if (mp < lowLimit) break;
        uint32_t const w = uni(prev[mp]);
        m = ZHIP_LZ_LINK(w);
        LZ_STAT(3, 1);
        if (w & ZHIP_HC_SKIPPED) continue;                                   // never inserted: it takes no slot of the row
        room--;
        if (uni((uint32_t)tags[mp]) != myTag) continue;
        attempts--;
        if (!done && uni(ld32(src + mp + ml - 3)) == uni(ld32(src + x + ml - 3))) {
            uint32_t const cur = (wave_count_fwd(src, x, mp, nm8));
            if (cur > ml) { ml = cur; off = x - mp; if (x + cur == bEnd) done = true; }
        }
    }
    mlOut = ml; offOut = off;
}

// ------------------------------------------------------------------ the live rows (row matcher, exact parse)
// What ZSTD_row_update (zstd_lazy.c:916-947) has put into the rows when the search at `upTo` starts: every position of [st.ins, upTo) that
// was not left out (flagged in prev[]; a job's hole), in order, 64 at a time.  A row keeps its 2^rowLog - 1 latest inserts (ZSTD_row_nextIndex
// :784-795 cycles through the slots 1 .. rowMask; slot 0 of the reference's tag row is the head), so insert i of a row goes to slot
// i mod (2^rowLog - 1): lanes of one row are ranked with ballots, the first reads the row's count, the last writes it back.
__device__ inline void lz_ring_catchup(const uint8_t* __restrict__ src, const ZhipUnit& u, const uint32_t* prev, LzState& st, uint32_t upTo)
{
    if (upTo <= st.ins) return;
    uint32_t const lane = (uint32_t)lane_id(), rowLog = u.rowLog, usable = (1u << rowLog) - 1u, keyBits = lz_key_bits(u);
    uint32_t qcN = st.ins + lane < upTo ? st.ins + lane : upTo - 1;
    uint32_t wN = prev[qcN]; uint64_t bN = ld64(src + qcN);                     // the next step's loads are in flight while this step's count makes its round trip
    for (uint32_t q0 = st.ins; q0 < upTo; q0 += 64) {
        uint32_t const q = q0 + lane, w = wN; uint64_t const bytes = bN;
        if (q0 + 64 < upTo) { qcN = q + 64 < upTo ? q + 64 : upTo - 1; wN = prev[qcN]; bN = ld64(src + qcN); }
        bool const in = q < upTo && !(w & ZHIP_HC_SKIPPED) && !(q >= st.holeStart && q < st.holeEnd);
        unsigned long long const inM = __ballot(in);
        LZ_STAT(4, 1);
        if (!inM) { LZ_STAT(5, 1); continue; }
        uint32_t tag; uint32_t const k = lz_key(bytes, u, tag);
        unsigned long long const same = wave_hash_group(k, keyBits) & inM;           // the step's inserted positions of my row
        uint32_t const rank = (uint32_t)__popcll(same & below_mask((int)lane)), total = (uint32_t)__popcll(same);
        uint32_t c = (in && rank == 0) ? st.ring.cnt[k] : 0u;
        c = __shfl(c, same ? first_lane(same) : 0);
        if (in) {
            if (rank + 1 == total) st.ring.cnt[k] = c + total;
            if (rank + usable >= total) {                                        // of more than a row's worth in one step only the latest stay
                size_t const at = ((size_t)k << rowLog) + (c + rank) % usable;
                st.ring.pos[at] = q; st.ring.tag[at] = (uint8_t)tag;
            }
        }
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
    st.ins = upTo;
}
// ZSTD_RowFindBestMatch (zstd_lazy.c:1141-1340) at x from the live rows: the row's entries most recent first (ZSTD_row_getMatchMask rotates by the
// head), those at or above lowLimit with x's tag, at most 2^min(searchLog, rowLog) of them; the longest wins, the earlier of equals (:1287
// `currentMl > ml`), a match that reaches the block's end ends the search.  Lane r holds the r-th most recent entry; every candidate lane
// measures its own match up to ZHIP_HC_CAP bytes, longer ones are measured wave-wide.  All results uniform.
// The last (<= 63) positions that are still to be inserted travel with the search — three dependent round trips in all: (1) their link words
// and bytes + x's bytes, (2) the counts of their rows + the count and every slot of x's row, as it was BEFORE them, (3) the candidates'
// bytes.  Those of them that fall into x's own row are its most recent entries: they go in front of what the slots held.
__device__ inline void lz_live_ring(const uint8_t* __restrict__ src, uint32_t bEnd, uint32_t x, const ZhipUnit& u, const uint32_t* prev, LzState& st, uint32_t lowLimit,
                                    uint32_t& mlOut, uint32_t& offOut)
{
    uint32_t const lane = (uint32_t)lane_id(), rowLog = u.rowLog, usable = (1u << rowLog) - 1u, nm8 = bEnd - 8, keyBits = lz_key_bits(u);
    uint32_t const capped = u.searchLog < rowLog ? u.searchLog : rowLog, attempts = 1u << capped;
    LzRing const R = st.ring;
    if (x - st.ins > 63) lz_ring_catchup(src, u, prev, st, x - 63);
    uint32_t const ins0 = st.ins, nPend = x - ins0;                             // lanes below nPend: a position to insert; the others (lane 63 always) look at x
    uint32_t const q = lane < nPend ? ins0 + lane : x;
    uint32_t const w = prev[q];
    uint32_t tagq; uint32_t const kq = lz_key(ld64(src + q), u, tagq);
    bool const in = lane < nPend && !(w & ZHIP_HC_SKIPPED) && !(q >= st.holeStart && q < st.holeEnd);
    uint32_t const kx = (uint32_t)__builtin_amdgcn_readlane(kq, 63), tag = (uint32_t)__builtin_amdgcn_readlane(tagq, 63);
    unsigned long long const inM = __ballot(in);
    unsigned long long const same = wave_hash_group(kq, keyBits) & inM;
    uint32_t const rank = (uint32_t)__popcll(same & below_mask((int)lane)), total = (uint32_t)__popcll(same);
    // round trip 2
    uint32_t cIns = (in && rank == 0) ? R.cnt[kq] : 0u;
    size_t const at = ((size_t)kx << rowLog) + (lane < usable ? lane : 0u);
    uint32_t const cw = R.cnt[kx], sp = R.pos[at], stg = R.tag[at];
    uint32_t const c = uni(cw), nOld = c < usable ? c : usable;
    cIns = __shfl(cIns, same ? first_lane(same) : 0);
    __threadfence_block();                                                      // the row of x HAS been read (the loads are complete): now the inserts may land in it
    __builtin_amdgcn_wave_barrier();
    if (in) {
        if (rank + 1 == total) R.cnt[kq] = cIns + total;
        if (rank + usable >= total) {
            size_t const ia = ((size_t)kq << rowLog) + (cIns + rank) % usable;
            R.pos[ia] = q; R.tag[ia] = (uint8_t)tagq;
        }
    }
    st.ins = x;
    // the candidates, most recent first: x's row among the pending positions (the higher the later), then the slots
    unsigned long long P = __ballot(in && kq == kx);
    uint32_t const nP = (uint32_t)__popcll(P);
    uint32_t const rr = lane - nP;
    bool have = lane >= nP && rr < nOld && lane < usable;
    uint32_t const from = have ? (c - 1u - rr) % usable : 0u;                   // the r-th most recent insert of the row is its number c - 1 - r
    uint32_t mp = __shfl(sp, (int)from), tg = __shfl(stg, (int)from);
    for (uint32_t i = 0; P != 0 && i < usable; i++) {
        int const l = 63 - __clzll((long long)P);
        P &= ~(1ull << l);
        uint32_t const tl = (uint32_t)__builtin_amdgcn_readlane(tagq, l);
        if (lane == i) { mp = ins0 + (uint32_t)l; tg = tl; have = true; }
    }
    bool cand = have && mp >= lowLimit && tg == tag;
    unsigned long long const cm = __ballot(cand);
    cand = cand && (uint32_t)__popcll(cm & below_mask((int)lane)) < attempts;
    uint32_t cur = 0;
    if (cand) {
        for (;;) {
            uint32_t const sameB = lane_same_fwd(src, x + cur, x - mp, nm8);
            cur += sameB;
            if (sameB < 8 || cur >= ZHIP_HC_CAP) break;
        }
    }
    unsigned long long rest = __ballot(cand);
    uint32_t ml = 3, off = 0;
    while (rest) {
        int const l = first_lane(rest);
        rest &= rest - 1;
        uint32_t len = (uint32_t)__builtin_amdgcn_readlane(cur, l);
        uint32_t const mpl = (uint32_t)__builtin_amdgcn_readlane(mp, l);
        if (len >= ZHIP_HC_CAP) len = uni(wave_count_fwd(src, x, mpl, nm8));
        if (len > ml) { ml = len; off = x - mpl; if (x + len == bEnd) break; }
    }
    __threadfence_block();
    mlOut = ml; offOut = off;
}

struct LzBlock {           // what is constant over one block
    const uint8_t* src; uint32_t bStart, bEnd, low, maxDist;
    uint32_t* prev; const uint8_t* tags; const LzRec* best;
};
__device__ __forceinline__ uint32_t lz_low_limit(const LzBlock& B, uint32_t x) { return (x - B.low > B.maxDist) ? x - B.maxDist : B.low; }   // internal.h:1312
__device__ __forceinline__ void lz_gap_rule(const LzBlock& B, const ZhipUnit& u, LzState& st, uint32_t x)      // zstd_lazy.c:916-947
{
    if (x > st.ntu && x - st.ntu > 384) {
        // (as rh_gap_rule, zhip_parse_lazy.h: a batch start whose first event is a repcode taken without a search leaves nextToUpdate where it was — the next batch start meets
        // the same gap, longer, and only flags what lies behind the part already flagged; a frame of short runs took 9 s per MiB at level 5 before)
        uint32_t f0 = st.ntu + 96;
        if (st.gapFlagged > f0) f0 = st.gapFlagged;
        if (x - 32 > f0) { lz_flag_range(B.src, u, B.prev, st, f0, x - 32); st.gapFlagged = x - 32; }
    }
}
// one ZSTD_HcFindBestMatch / ZSTD_RowFindBestMatch call of the reference at x: insertion bookkeeping + the (pre)computed result
__device__ inline void lz_search(const LzBlock& B, const ZhipUnit& u, LzState& st, uint32_t x, const LzRec& rec, uint32_t& ml, uint32_t& offBase)
{
    if (u.rowLog) {
        if (!st.skipping) lz_gap_rule(B, u, st, x);
        else if (st.ntu < x) lz_flag_range(B.src, u, B.prev, st, st.ntu, x);
        st.ntu = x + 1;
    } else {
        if (st.skipping && st.ntu + 1 < x) lz_flag_range(B.src, u, B.prev, st, st.ntu + 1, x);      // :651 only nextToUpdate itself is inserted
        st.ntu = x;
    }
    lz_reconcile(B.src, u, B.prev, st, x);               // everything below x is decided now: was it what the predicting parse expected?
    uint32_t off;
    if (!st.predict) LZ_STAT(0, 1);
    bool live = rec.mode == 3 || x + ZHIP_HC_CAP > B.bEnd;
    if (!live && st.gapEnd != 0 && rec.minCand != ZHIP_LZ_NONE && rec.minCand <= st.gapEnd) live = lz_dirty(B.src, u, st, x);
    if (live) {
        uint32_t const lowLimit = lz_low_limit(B, x);
        if (u.rowLog && st.ring.cnt && st.nFlagged && !st.predict) {
            LZ_STAT(1, 1);
            lz_live_ring(B.src, B.bEnd, x, u, B.prev, st, lowLimit, ml, off);
        }
        else if (u.rowLog) { if (!st.predict) LZ_STAT(2, 1); lz_live_rh(B.src, B.bEnd, x, B.prev, B.tags, u.searchLog, u.rowLog, lowLimit, ml, off); }
        else lz_live_hc(B.src, B.bEnd, x, B.prev, u.searchLog, u.chainLog, lowLimit, ml, off);
    }
    else if (rec.mode == 0) { ml = rec.b; off = rec.a; }
    else {
        uint32_t const nm8 = B.bEnd - 8;
        ml = (wave_count_fwd(B.src, x, rec.a, nm8)); off = x - rec.a;
        if (rec.mode == 2 && x + ml != B.bEnd) {
            uint32_t const l2 = (wave_count_fwd(B.src, x, rec.b, nm8));
            if (l2 > ml) { ml = l2; off = x - rec.b; }
        }
    }
    offBase = off + 3;
}

// ZSTD_compressBlock_lazy_generic (zstd_lazy.c:1516-1779, noDict) over the block [bStart, bEnd) of a window whose match state lives in
// prev / tags / st; B.low = window.lowLimit after ZSTD_window_enforceMaxDist(block start)
__device__ inline void parse_lazy_block(const LzBlock& B, const ZhipUnit& u, LzState& st, uint32_t rep1, uint32_t rep2, uint32_t rep3,
                                        ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    const uint8_t* const src = B.src;
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const depth = (uint32_t)u.strategy - 3;
    uint32_t const bStart = B.bStart, bEnd = B.bEnd, bLen = bEnd - bStart, guard = u.rowLog ? 16u : 8u;     // :1527
    FastOut out; out.seqs = seqs; out.lits = lits; out.nbSeq = 0; out.longPos = 0; out.longType = 0;
    out.litPos = 0; out.pendV = 0; out.pendSh = 0; out.pendOff = 0; out.pendLen = 0;
    uint32_t anchor = bStart, off1 = rep1, off2 = rep2, saved1 = 0, saved2 = 0;
    // zstd_compress.c:3243-3249 limited update after a very long match; zstd_lazy.c:1567
    if (bStart > st.ntu + 384) {
        uint32_t const d = bStart - st.ntu - 384, nn = bStart - (d < 192 ? d : 192);
        if (u.rowLog && st.gapFlagged > nn) {
            // lz_gap_rule flags at a batch START, for the search the batch opens with — but greedy takes a repcode without one, and when no search came before the block
            // ended, what was flagged behind the new nextToUpdate is inserted after all (the reference's next search starts at nn): taken back, their keys marked like any
            // position decided otherwise than expected.  (Round 6: found by tests/tools/gpu_fuzz_shapes.py — runs of 24 with a byte flipped every 97, level 5, 400 KB.)
            uint32_t const g1 = st.gapFlagged;
            for (uint32_t q0 = nn; q0 < g1; q0 += 64) {
                uint32_t const q = q0 + (uint32_t)lane_id();
                if (q < g1) {
                    if (st.predict) B.prev[q] &= ~ZHIP_LZ_PRED;
                    else { B.prev[q] &= ~ZHIP_HC_SKIPPED; uint32_t tag; uint32_t const k = lz_key(ld64(src + q), u, tag); atomicOr(&st.dirty[k >> 5], 1u << (k & 31)); }
                }
            }
            __threadfence_block();
            __builtin_amdgcn_wave_barrier();
            if (st.predict) st.nPred -= g1 - nn;
            else { if (g1 - 1 > st.gapEnd) st.gapEnd = g1 - 1; st.epoch++; }
        }
        lz_flag_range(src, u, B.prev, st, st.ntu, nn);
        st.ntu = nn;
    }
    st.gapFlagged = 0;
    st.skipping = 0;
    if (bLen > guard) {
        uint32_t const nm8 = bEnd - 8, ilimit = bEnd - guard;
        uint32_t const rowBias = u.rowLog ? 1u : 0u;
        uint32_t ip = bStart + (bStart == B.low ? 1u : 0u);                    // :1552
        {   uint32_t const wl = lz_low_limit(B, ip), maxRep = ip - wl;       // :1553-1559
            if (off2 > maxRep) { saved2 = off2; off2 = 0; }
            if (off1 > maxRep) { saved1 = off1; off1 = 0; }
        }
        while (ip < ilimit) {                                                // :1581
            uint32_t const step = ((ip - anchor) >> 8) + 1;                  // :1614 kSearchStrength = 8
            uint32_t x, K = 0, ip0 = ip;
            LzRec recj; recj.a = 0; recj.b = 0; recj.minCand = ZHIP_LZ_NONE; recj.mode = 0;
            bool repj = false;
            bool repHit; LzRec rec;
            unsigned long long ev = 0;
            if (step <= 8) {
                if (u.rowLog) lz_gap_rule(B, u, st, ip);                     // the batch's first search (not lazy-skipping) meets the gap since nextToUpdate
                uint32_t const xj = ip + lane * step;
                bool const valid = xj < ilimit && ((xj - anchor) >> 8) + 1 == step;
                uint32_t const xc = xj < nm8 ? xj : nm8;
                if (valid) recj = B.best[xj];
                uint32_t const cur4 = ld32(src + xc + 1), rv = ld32(src + (xc + 1 - off1));
                repj = valid && off1 > 0 && rv == cur4;                      // :1600 repcode at ip+1
                if (!st.predict && st.havePred) {                            // the batch's positions (and what lies between them) are inserted by its searches
                    uint32_t const Kv = (uint32_t)__popcll(__ballot(valid));
                    if (Kv) lz_reconcile(src, u, B.prev, st, ip + (Kv - 1) * step);
                }
                bool stale = valid && (xj + ZHIP_HC_CAP > bEnd);
                if (!stale && valid && st.gapEnd != 0 && recj.minCand != ZHIP_LZ_NONE && recj.minCand <= st.gapEnd) stale = lz_dirty(src, u, st, xc);
                bool const needLive = valid && (recj.mode == 3 || stale);
                bool const found = valid && (recj.mode != 0 || recj.b >= 4);
                K = (uint32_t)__popcll(__ballot(valid));
                ev = __ballot(repj || needLive || found);
                if (!ev) {                                                   // K failed searches (:1613-1624), lazySkipping = 0
                    st.ntu = ip + (K - 1) * step + rowBias; st.skipping = 0;
                    ip = ip + K * step;
                    continue;
                }
            }
            uint32_t const epoch1 = st.epoch;                               // what the batch found out about its records' staleness holds while this does not move
            uint32_t matchLength = 0, start = 0, offBase = 1;
            bool direct = false, failed = false;
            // the events of the batch in order: a search that fails (no match, no repcode) moves on to the batch's next event without
            // loading the batch again — on long-match data without the prediction nearly every search is such a live search that fails
            for (;;) {
                int e = 0;
                if (step <= 8) {
                    e = first_lane(ev);
                    x = ip0 + (uint32_t)e * step;
                    if (e > 0) { st.ntu = x - step + rowBias; st.skipping = 0; }
                    repHit = (__ballot(repj) >> e) & 1;
                    rec = lz_rec_lane(recj, e);
                } else {
                    x = ip;
                    LzRec const r0 = B.best[x];
                    rec = lz_rec_lane(r0, 0);
                    repHit = off1 > 0 && uni(ld32(src + x + 1)) == uni(ld32(src + (x + 1 - off1)));
                }
                matchLength = 0; start = x + 1; offBase = 1; direct = false;
                if (repHit) {                                                // :1600-1604
                    matchLength = 4 + (wave_count_fwd(src, x + 5, x + 5 - off1, nm8));
                    if (depth == 0) direct = true;
                }
                ip = x;
                if (direct) break;
                {   uint32_t ml2, ob2;                                       // :1607-1611
                    lz_search(B, u, st, x, rec, ml2, ob2);
                    if (ml2 > matchLength) { matchLength = ml2; start = x; offBase = ob2; }
                }
                if (matchLength >= 4) break;
                failed = true;                                               // :1613-1625
                if (step <= 8 && st.epoch == epoch1) {
                    ev &= ~below_mask(e + 1);
                    if (ev) { failed = false; continue; }
                    st.ntu = ip0 + (K - 1) * step + rowBias; st.skipping = 0;   // the batch's remaining searches fail from their records
                    ip = ip0 + K * step;
                    break;
                }
                uint32_t const stp = ((x - anchor) >> 8) + 1;
                ip = x + stp;
                st.skipping = stp > 8;                                       // kLazySkippingStep = 8
                break;
            }
            if (failed) continue;
            bool const window = step == 1;
            auto rec_at = [&](uint32_t q) -> LzRec {
                if (window && q - ip0 < K) return lz_rec_lane(recj, (int)(q - ip0));
                LzRec const r = B.best[q];
                return lz_rec_lane(r, 0);
            };
            auto rep_at = [&](uint32_t q) -> bool {           // MEM_read32(q) == MEM_read32(q - off1), off1 > 0
                if (window && q - 1 - ip0 < K) return (__ballot(repj) >> (q - 1 - ip0)) & 1;
                return uni(ld32(src + q)) == uni(ld32(src + (q - off1)));
            };
            if (!direct) {
                if (depth >= 1) {
                    while (ip < ilimit) {                                    // :1628-1700
                        ip++;
                        if (off1 > 0 && rep_at(ip)) {
                            uint32_t const mlRep = 4 + (wave_count_fwd(src, ip + 4, ip + 4 - off1, nm8));
                            int const gain2 = (int)(mlRep * 3);
                            int const gain1 = (int)(matchLength * 3 - hb32(offBase) + 1);
                            if (gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                        }
                        {   uint32_t ml2, ob2;
                            lz_search(B, u, st, ip, rec_at(ip), ml2, ob2);
                            int const gain2 = (int)(ml2 * 4 - hb32(ob2));
                            int const gain1 = (int)(matchLength * 4 - hb32(offBase) + 4);
                            if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = ob2; start = ip; continue; }
                        }
                        if (depth == 2 && ip < ilimit) {                     // :1663-1698
                            ip++;
                            if (off1 > 0 && rep_at(ip)) {
                                uint32_t const mlRep = 4 + (wave_count_fwd(src, ip + 4, ip + 4 - off1, nm8));
                                int const gain2 = (int)(mlRep * 4);
                                int const gain1 = (int)(matchLength * 4 - hb32(offBase) + 1);
                                if (gain2 > gain1) { matchLength = mlRep; offBase = 1; start = ip; }
                            }
                            {   uint32_t ml2, ob2;
                                lz_search(B, u, st, ip, rec_at(ip), ml2, ob2);
                                int const gain2 = (int)(ml2 * 4 - hb32(ob2));
                                int const gain1 = (int)(matchLength * 4 - hb32(offBase) + 7);
                                if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offBase = ob2; start = ip; continue; }
                            }
                        }
                        break;
                    }
                }
                if (offBase > 3) {                                           // :1707-1714 catch up, the match stays above the window's low end
                    uint32_t const off = offBase - 3, match = start - off;
                    uint32_t const lim = (start - anchor) < (match - B.low) ? (start - anchor) : (match - B.low);
                    uint32_t const back = (wave_count_back(src, start, match, lim));
                    start -= back; matchLength += back;
                    off2 = off1; off1 = off;
                }
            }
            if (!st.predict) {
            lits_copy(out, src, nm8, anchor, start - anchor);                // :1727-1731
            store_seq(out, start - anchor, offBase, matchLength);
            }
            anchor = ip = start + matchLength;
            st.skipping = 0;                                                 // :1732-1738
            while (ip <= ilimit && off2 > 0) {                               // :1763-1773
                if (uni(ld32(src + ip)) != uni(ld32(src + (ip - off2)))) break;
                uint32_t const rl = 4 + (wave_count_fwd(src, ip + 4, ip + 4 - off2, nm8));
                {   uint32_t const t = off2; off2 = off1; off1 = t; }
                if (!st.predict) store_seq(out, 0, 1, rl);
                ip += rl; anchor = ip;
            }
        }
        if (!st.predict) {
        lits_copy(out, src, nm8, anchor, bEnd - anchor);                    // trailing literals (zstd_compress.c:3365)
        lits_flush(out);
        }
    } else if (!st.predict) {
        for (uint32_t i = lane; i < bLen; i += 64) lits[i] = src[bStart + i];
        out.litPos = bLen;
    }
    saved2 = (saved1 != 0 && off1 != 0) ? saved1 : saved2;                  // :1777-1783
    if (lane == 0) {
        meta->nbSeq = out.nbSeq; meta->lastLits = bEnd - anchor;
        meta->longPos = out.longPos; meta->longType = out.longType;
        meta->rep[0] = off1 ? off1 : saved1; meta->rep[1] = off2 ? off2 : saved2; meta->rep[2] = rep3;
        meta->status = 0; meta->litSize = out.litPos; meta->pad0 = 0;
    }
}

// ------------------------------------------------------------------ the block splitter of ZSTD_lazy2
// ZSTD_splitBlock(split_lvl1) = ZSTD_splitBlock_byChunks with one 2-byte event out of five (zstd_preSplit.c:47-58, :79-114, :141-181):
// the first 8 KB chunk of the next 128 KB whose fingerprint is "too different" from what came before it starts the next block.
struct LzSplitShared { uint32_t past[1024]; uint32_t cur[1024]; uint32_t dev; uint32_t pad[3]; };
__device__ inline uint32_t lz_split_lvl1(const uint8_t* __restrict__ p, LzSplitShared* sp)
{
    uint32_t const t = threadIdx.x;
    uint32_t const limit = 8192 - 2 + 1, nEv = limit / 5;                    // what nbEvents grows by per chunk (:79-83)
    for (uint32_t i = t; i < 1024; i += ZHIP_ENT_THREADS) sp->past[i] = 0;
    __syncthreads();
    for (uint32_t i = 5u * t; i < limit; i += 5u * ZHIP_ENT_THREADS) {
        uint32_t const v = (uint32_t)p[i] | ((uint32_t)p[i + 1] << 8);
        atomicAdd(&sp->past[(v * 0x9e3779b9u) >> 22], 1u);
    }
    uint32_t pastN = nEv, penalty = 3, result = ZHIP_UNIT_MAX;
    for (uint32_t pos = 8192; pos <= ZHIP_UNIT_MAX - 8192; pos += 8192) {
        for (uint32_t i = t; i < 1024; i += ZHIP_ENT_THREADS) sp->cur[i] = 0;
        if (t == 0) sp->dev = 0;
        __syncthreads();
        for (uint32_t i = 5u * t; i < limit; i += 5u * ZHIP_ENT_THREADS) {
            uint32_t const v = (uint32_t)p[pos + i] | ((uint32_t)p[pos + i + 1] << 8);
            atomicAdd(&sp->cur[(v * 0x9e3779b9u) >> 22], 1u);
        }
        __syncthreads();
        uint32_t mine = 0;                                                   // :87-97 fpDistance (every term and the sum fit 32 bits: <= 2 * 24585 * 1638)
        for (uint32_t k = t; k < 1024; k += ZHIP_ENT_THREADS) {
            int32_t const d = (int32_t)(sp->past[k] * nEv) - (int32_t)(sp->cur[k] * pastN);
            mine += (uint32_t)(d < 0 ? -d : d);
        }
        atomicAdd(&sp->dev, mine);
        __syncthreads();
        uint32_t const deviation = __builtin_amdgcn_readfirstlane(sp->dev);      // one value for the workgroup: the loop's exit is uniform (scripts/scan_divergent_barriers.py)
        uint32_t const threshold = (uint32_t)(((uint64_t)pastN * nEv * (14u + penalty)) / 16u);      // :102-114
        if (deviation >= threshold) { result = pos; break; }                 // uniform: every thread read the same sum
        for (uint32_t k = t; k < 1024; k += ZHIP_ENT_THREADS) sp->past[k] += sp->cur[k];
        pastN += nEv;
        if (penalty > 0) penalty--;
        __syncthreads();
    }
    __syncthreads();
    return result;
}

// ------------------------------------------------------------------ the frame / job loop
struct LzFrameShared {
    ZhipParse meta;
    uint32_t flag;
    uint32_t pad[5];
    LzSplitShared split;
};
__host__ __device__ inline uint32_t frame_lazy_lds_bytes()
{
    return (uint32_t)((sizeof(EntShared) + 15) & ~(size_t)15) + (uint32_t)sizeof(LzFrameShared);
}

// ZSTD_fseBitCost (zstd_compress_sequences.c:103-135) is in zhip_entropy.h (fse_bit_cost_wave); the carried table marks the symbols it
// does not cover with the transform of a zero-probability symbol, whose cost is the "bad" cost
__device__ inline void lz_keep_fse_table(ZhipDictEntropy* ent, const EntShared* sh, int k)
{
    uint32_t const t = threadIdx.x;
    const uint32_t* const from = (const uint32_t*)&sh->ct[k];
    uint32_t* const to = (uint32_t*)&ent->ct[k];
    for (uint32_t i = t; i < sizeof(FseCTable) / 4; i += ZHIP_ENT_THREADS) to[i] = from[i];
    __syncthreads();
    uint32_t const tl = sh->ct[k].tableLog, maxSym = sh->maxCode[k];
    if (t > maxSym && t < 56) ent->ct[k].dBits[t] = ((tl + 1) << 16) - (1u << tl);
}

// The PREDICTING parse of a window (one wavefront; see rh_reconcile in zhip_parse_lazy.h): the block loop with fixed block sizes (the real borders depend on
// the compressed sizes; a border a few KB off only costs a few mispredictions), every block taken as confirmed, nothing stored.  Data without long matches
// leaves almost nothing un-inserted: the first 32 KB are a probe — when they marked less than 1 position in 64 the marks are taken back and the window is
// parsed once, as if there were no prediction (returns false: the exact parse then has nothing to compare with).
__device__ inline bool frame_lazy_predict(const uint8_t* __restrict__ src, const ZhipUnit& u, uint32_t* prev, const uint8_t* tags, const LzRec* best,
                                          ZhipParse* meta /* LDS */, const ZhipJob* __restrict__ job)
{
    bool const first = !job || (job->flags & ZHIP_JOB_FIRST);
    uint32_t const j0 = job ? job->prefixLen : 0u, jEnd = j0 + u.srcLen, maxDist = 1u << u.windowLog;
    uint32_t rep1 = first ? 1u : 0u, rep2 = first ? 4u : 0u, rep3 = first ? 8u : 0u, low = 0;
    LzState ls; ls.ntu = j0; ls.skipping = 0; ls.gapEnd = 0; ls.gapFlagged = 0; ls.dirty = nullptr; ls.predict = 1; ls.scanned = j0; ls.nPred = 0; ls.havePred = 0;
    ls.ring.cnt = nullptr; ls.ring.pos = nullptr; ls.ring.tag = nullptr; ls.ins = 0; ls.nFlagged = 0; ls.epoch = 0; ls.holeStart = 0; ls.holeEnd = 0;
    for (uint32_t pos = j0; pos < jEnd; ) {
        // the 32 KB probe, then the rest of the first 128 KB block, then whole blocks: the predicted block borders stay on the exact parse's j0 + k * 128 KB
        uint32_t const most = pos == j0 ? 32768u : (pos == j0 + 32768u ? ZHIP_UNIT_MAX - 32768u : ZHIP_UNIT_MAX);
        uint32_t const bLen = jEnd - pos < most ? jEnd - pos : most;
        if (bLen >= 7) {
            if (pos > maxDist && pos - maxDist > low) low = pos - maxDist;
            LzBlock B; B.src = src; B.bStart = pos; B.bEnd = pos + bLen; B.low = low; B.maxDist = maxDist; B.prev = prev; B.tags = tags; B.best = best;
            parse_lazy_block(B, u, ls, rep1, rep2, rep3, nullptr, nullptr, meta);
            __builtin_amdgcn_wave_barrier();
            rep1 = meta->rep[0]; rep2 = meta->rep[1]; rep3 = meta->rep[2];
            __builtin_amdgcn_wave_barrier();
        }
        if (pos == j0 && ls.nPred * 64u < bLen) {
            if (ls.nPred) {
                for (uint32_t q = pos + (uint32_t)lane_id(); q < pos + bLen; q += 64) { uint32_t const w = prev[q]; if (w & ZHIP_LZ_PRED) prev[q] = w & ~ZHIP_LZ_PRED; }
                __threadfence_block();
            }
            return false;
        }
        pos += bLen;
    }
    return true;
}

// job == nullptr: the whole input src[0, u.srcLen) as one frame; else one job of a frame, src = the start of the job's window
__device__ inline void frame_lazy(const uint8_t* __restrict__ src, const ZhipUnit& u, const ZhipLzSlot& L, uint32_t* prev, const uint8_t* tags, const LzRec* best,
                                  uint32_t* dirty, uint8_t* ring /* lz_ring_bytes(u), or nullptr */, ZhipSeq* seqs, uint8_t* lits, uint16_t* stBits, uint32_t seqCap, uint8_t* __restrict__ out, uint32_t* outSize,
                                  EntShared* sh, LzFrameShared* fs, ZhipFrameState* st, bool withChecksum, uint32_t checksum, const ZhipJob* __restrict__ job, bool havePred)
{
    int const t = (int)threadIdx.x, wv = t >> 6;
    bool const first = !job || (job->flags & ZHIP_JOB_FIRST), lastJob = !job || (job->flags & ZHIP_JOB_LAST);
    uint32_t const j0 = job ? job->prefixLen : 0u;
    uint32_t const n = u.srcLen, jEnd = j0 + n;
    uint32_t const frameSize = job ? (uint32_t)job->frameSize : n;
    uint32_t op = first ? frame_header_bytes_multi(frameSize, u.windowLog) : 0u;
    if (first && t == 0) write_frame_header_multi(out, frameSize, u.windowLog, withChecksum);
    if (n == 0) {
        if (t == 0) {
            out[op] = 1; out[op + 1] = 0; out[op + 2] = 0; op += 3;
            if (withChecksum) { for (int b = 0; b < 4; b++) out[op + b] = (uint8_t)(checksum >> (8 * b)); op += 4; }
            *outSize = op;
        }
        return;
    }
    {   uint32_t const words = ((1u << lz_key_bits(u)) + 31u) >> 5;         // the dirty-key bitmap
        for (uint32_t i = (uint32_t)t; i < words; i += ZHIP_ENT_THREADS) dirty[i] = 0;
    }
    if (t == 0) { st->ent.hufRepeat = 0; st->ent.hufMaxSym = 0; st->ent.fseRepeat[0] = 0; st->ent.fseRepeat[1] = 0; st->ent.fseRepeat[2] = 0; }
    uint32_t rep1 = first ? 1u : 0u, rep2 = first ? 4u : 0u, rep3 = first ? 8u : 0u;     // later jobs: ZSTD_invalidateRepCodes (zstdmt_compress.c:741)
    long long savings = (job && !first) ? -(long long)job->ownHeader : 0;
    uint32_t pos = j0, low = 0;
    uint32_t const maxDist = 1u << u.windowLog;
    LzState ls; ls.ntu = j0; ls.skipping = 0; ls.gapEnd = 0; ls.gapFlagged = 0; ls.dirty = dirty; ls.predict = 0; ls.scanned = j0; ls.nPred = 0; ls.havePred = (havePred && st->predicted) ? 1u : 0u;      // a job: nextToUpdate = the prefix's end
    ls.ring.cnt = nullptr; ls.ring.pos = nullptr; ls.ring.tag = nullptr; ls.ins = L.linkStart; ls.nFlagged = 0; ls.epoch = 0; ls.holeStart = L.holeStart; ls.holeEnd = L.holeEnd;
    if (ring && u.rowLog) {                                                  // fresh rows: every count 0 (the slots are only read below a count)
        uint32_t const rows = 1u << lz_key_bits(u);
        ls.ring.cnt = (uint32_t*)ring; ls.ring.pos = ls.ring.cnt + rows; ls.ring.tag = (uint8_t*)(ls.ring.pos + ((size_t)rows << u.rowLog));
        for (uint32_t i = (uint32_t)t; i < rows; i += ZHIP_ENT_THREADS) ls.ring.cnt[i] = 0;
    }
    __threadfence_block();
    __syncthreads();
    while (pos < jEnd) {
        uint32_t const chunkEnd = job ? (jEnd - pos > ZHIP_JOB_CHUNK - ((pos - j0) & (ZHIP_JOB_CHUNK - 1)) ? pos + ZHIP_JOB_CHUNK - ((pos - j0) & (ZHIP_JOB_CHUNK - 1)) : jEnd) : jEnd;
        uint32_t const remaining = chunkEnd - pos;
        uint32_t bLen = remaining < ZHIP_UNIT_MAX ? remaining : ZHIP_UNIT_MAX;                      // zstd_compress.c:4494-4518
        if (remaining >= ZHIP_UNIT_MAX && savings >= 3) {
            if (u.strategy >= ZHIP_STRAT_LAZY2) bLen = remaining <= ZHIP_UNIT_MAX ? remaining : lz_split_lvl1(src + pos, &fs->split);
            else bLen = ZHIP_FRAME_BLOCK_SPLIT;
        }
        uint32_t const last = (lastJob && pos + bLen == jEnd) ? 1u : 0u;
        uint32_t const end = pos + bLen;
        uint8_t* const body = out + op + 3;
        uint32_t cSize = 0;
        if (bLen >= 7) {                                                                         // :3216
            if (pos > maxDist && pos - maxDist > low) low = pos - maxDist;                       // :4561 ZSTD_window_enforceMaxDist(block start)
            if (wv == 0) {
                LzBlock B; B.src = src; B.bStart = pos; B.bEnd = end; B.low = low; B.maxDist = maxDist; B.prev = prev; B.tags = tags; B.best = best;
                parse_lazy_block(B, u, ls, rep1, rep2, rep3, seqs, lits, &fs->meta);
            }
            __threadfence_block();
            __syncthreads();
            ZhipParse const pm = fs->meta;
            cSize = entropy_block<ZHIP_ENT_THREADS, EntShared>(src + pos, bLen, u, seqs, pm, lits, stBits, seqCap, body, sh, &st->ent);
            if (pos != j0 && cSize < 25) {                                   // :4365-4376 an RLE block, never the context's first one
                if (t == 0) fs->flag = 0;
                __syncthreads();
                uint8_t const b0 = src[pos];
                bool diff = false;
                for (uint32_t i = (uint32_t)t; i < bLen; i += ZHIP_ENT_THREADS) diff = diff || src[pos + i] != b0;
                if (diff) fs->flag = 1;
                __syncthreads();
                if (fs->flag == 0) cSize = 1;
            }
            if (cSize > 1) {                                                 // :4379-4381 the block confirms repcodes and tables
                rep1 = pm.rep[0]; rep2 = pm.rep[1]; rep3 = pm.rep[2];
                if (sh->litMode == 2 && sh->litType == 2) {
                    st->ent.hufCode[t] = sh->code[t];
                    if (t == 0) { st->ent.hufRepeat = 1; st->ent.hufMaxSym = sh->hufMaxSym; }
                }
                if (pm.nbSeq > 0) {                                          // the tables the sequences were coded with, and their states (zstd_compress_sequences.c:157-235)
                    for (int k = 0; k < 3; k++) {
                        uint32_t const ty = sh->encType[k];
                        if (ty == 2) { lz_keep_fse_table(&st->ent, sh, k); if (t == 0) st->ent.fseRepeat[k] = 1; }
                        else if (ty != 3 && t == 0) st->ent.fseRepeat[k] = 0;
                    }
                }
            }
        }
        uint32_t total;
        if (cSize == 0) {                                                    // :4592 ZSTD_noCompressBlock
            __syncthreads();
            for (uint32_t i = (uint32_t)t; i < bLen; i += ZHIP_ENT_THREADS) body[i] = src[pos + i];
            if (t == 0) { uint32_t const bh = last + (0u << 1) + (bLen << 3); out[op] = (uint8_t)bh; out[op + 1] = (uint8_t)(bh >> 8); out[op + 2] = (uint8_t)(bh >> 16); }
            total = 3 + bLen;
        } else if (cSize == 1) {
            __syncthreads();
            if (t == 0) { uint32_t const bh = last + (1u << 1) + (bLen << 3); out[op] = (uint8_t)bh; out[op + 1] = (uint8_t)(bh >> 8); out[op + 2] = (uint8_t)(bh >> 16); body[0] = src[pos]; }
            total = 4;
        } else {
            if (t == 0) { uint32_t const bh = last + (2u << 1) + (cSize << 3); out[op] = (uint8_t)bh; out[op + 1] = (uint8_t)(bh >> 8); out[op + 2] = (uint8_t)(bh >> 16); }
            total = 3 + cSize;
        }
        op += total;
        savings += (long long)bLen - (long long)total;
        if (job && first && end == chunkEnd && end - j0 <= ZHIP_JOB_CHUNK) savings -= (long long)frame_header_bytes_multi(frameSize, u.windowLog);
        pos = end;
        __threadfence_block();
        __syncthreads();
    }
    if (t == 0) {
        if (withChecksum && lastJob) { for (int b = 0; b < 4; b++) out[op + b] = (uint8_t)(checksum >> (8 * b)); op += 4; }
        *outSize = op;
    }
}

}  // namespace zhip
