// zhip_tables.h — entropy *table construction* for the gfx950 entropy kernel, by whole wavefronts.
//
// WHAT (outputs are the reference's, bit for bit): FSE normalised counts (lib/compress/fse_compress.c:379-525), the
// NCount table description (:234-327), the FSE encoding table (:68-214), the length-limited Huffman code
// (lib/compress/huf_compress.c:376-498, :620-791) and its tree description with FSE-compressed weights (:147-289).
//
// HOW (CDNA4; the shape is ours): an alphabet has at most 64 (FSE) or 256 (Huffman) symbols, so one lane owns one
// symbol and every table is built by wave-wide reductions, scans and ballots over LDS-resident arrays:
//   * normalisation  — one lane per symbol, wave sum / first-maximum; the secondary distribution is three classification
//                      ballots and one 64-bit prefix scan;
//   * NCount         — the bit-field of a symbol depends only on the prefix sum of |norm| before it (that fixes the
//                      remaining budget, hence threshold and field width) and on the zero run in front of it: every lane
//                      computes its own field, a scan of the widths gives the bit positions, the fields are OR-ed in;
//   * encoding table — the reference's symbol spread visits cells (j * step) mod size and skips the low-probability area:
//                      64 cells per round get their rank by ballot + popcount and their symbol by a binary search in
//                      the prefix sums; the per-symbol state numbering is a stable grouping by ballot rounds;
//   * Huffman        — the order-by-count is a rank computed from all-pairs key comparisons (bucket, symbol); the tree
//                      depths come from parent links by chunked relaxation; canonical values from per-length ballot
//                      groups.  Only what DEFINES the reference's tie-breaking stays serial on one lane: the quicksort
//                      inside a log2 bucket (unstable, huf_compress.c:571-607), the two-queue merge (:681-718) and the
//                      height repair (:376-498).
// Every function below is called by ALL 64 lanes of one wavefront with wave-uniform arguments; results are wave-uniform.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zhip {

__host__ __device__ __forceinline__ uint32_t hb32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }   // lib/common/bits.h:177

// ------------------------------------------------------------------ shared table types
struct FseCTable {            // encoder view of one FSE table (lib/common/fse.h:437-476 FSE_symbolCompressionTransform)
    uint16_t state[512];      // next-state table, sorted by symbol
    int32_t  dFind[56];       // deltaFindState
    uint32_t dBits[56];       // deltaNbBits
    uint32_t tableLog;
};

// The entropy state a ZDICT-format dictionary gives the first block of every frame (ZSTD_loadCEntropy,
// lib/compress/zstd_compress.c:4986-5076): built once on the host (zhip_cdict_host.h), read by k_entropy.
// repeat modes: 0 none, 1 "check" (usable only if it covers the block's symbols), 2 "valid" (covers every symbol).
struct ZhipDictEntropy {
    uint32_t hufRepeat, hufMaxSym;
    uint32_t hufCode[256];        // value << 8 | nbBits, as EntShared::code
    uint32_t fseRepeat[3];        // LL, OF, ML
    FseCTable ct[3];              // LL, OF, ML
};

// table log for `n` symbols drawn from an alphabet whose largest symbol is maxSym (fse_compress.c:348-369)
__host__ __device__ inline uint32_t fse_min_table_log(uint32_t n, uint32_t maxSym)
{
    uint32_t const bySize = hb32(n) + 1, byAlphabet = hb32(maxSym) + 2;
    return bySize < byAlphabet ? bySize : byAlphabet;
}
__host__ __device__ inline uint32_t fse_optimal_table_log(uint32_t maxLog, uint32_t n, uint32_t maxSym, uint32_t minus)
{
    uint32_t log = maxLog;
    uint32_t const bySrc = hb32(n - 1) - minus, floorLog = fse_min_table_log(n, maxSym);
    if (bySrc < log) log = bySrc;
    if (floorLog > log) log = floorLog;
    return log < 5 ? 5 : (log > 12 ? 12 : log);
}

// per-symbol transform of an encoding table (fse.h:437-450): what a symbol of normalised count nv adds to a state
__host__ __device__ __forceinline__ void fse_symbol_transform(uint32_t tableLog, int nv, uint32_t before /* cells of the symbols in front */, uint32_t* dBits, int32_t* dFind)
{
    if (nv == 0) { *dBits = ((tableLog + 1) << 16) - (1u << tableLog); *dFind = 0; return; }
    uint32_t const occ = nv < 0 ? 1u : (uint32_t)nv;                    // a low-probability symbol owns one cell
    uint32_t const maxBitsOut = occ == 1 ? tableLog : tableLog - hb32(occ - 1);
    *dBits = (maxBitsOut << 16) - (occ << maxBitsOut);
    *dFind = (int32_t)(before - occ);
}

#if !defined(ZHIP_TABLES_HOST_ONLY)
// ------------------------------------------------------------------ wave primitives (all 64 lanes active)
__device__ __forceinline__ uint32_t tw_lane() { return threadIdx.x & 63u; }
__device__ __forceinline__ unsigned long long tw_below(uint32_t l) { return l >= 64 ? ~0ull : ((1ull << l) - 1); }
__device__ __forceinline__ uint32_t tw_incl(uint32_t v)
{
    uint32_t const lane = tw_lane();
    for (uint32_t d = 1; d < 64; d <<= 1) { uint32_t const o = __shfl_up(v, d); if (lane >= d) v += o; }
    return v;
}
__device__ __forceinline__ uint64_t tw_incl64(uint64_t v)
{
    uint32_t const lane = tw_lane();
    for (uint32_t d = 1; d < 64; d <<= 1) {
        uint32_t const lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
        if (lane >= d) v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ uint32_t tw_sum(uint32_t v) { return __builtin_amdgcn_readlane(tw_incl(v), 63); }
__device__ __forceinline__ uint32_t tw_max(uint32_t v)
{
    for (int m = 32; m; m >>= 1) { uint32_t const o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}
// lowest lane among `cand` whose value equals the maximum over `cand` (64 when cand is empty)
__device__ __forceinline__ uint32_t tw_first_max(uint32_t v, bool cand, uint32_t* maxOut)
{
    uint32_t const m = tw_max(cand ? v : 0u);
    unsigned long long const at = __ballot(cand && v == m);
    *maxOut = m;
    return at ? (uint32_t)(__ffsll((long long)at) - 1) : 64u;
}

// ------------------------------------------------------------------ FSE normalisation
// secondary distribution (fse_compress.c:379-463): lane = symbol, `n` holds the lane's result
__device__ inline bool fse_spread_rest_wave(int& n, uint32_t c, bool on, uint32_t tableLog, uint32_t total, uint32_t maxSym, int lowProb)
{
    uint32_t const lane = tw_lane();
    uint32_t const tiny = total >> tableLog;
    uint32_t one = (uint32_t)(((uint64_t)total * 3) >> (tableLog + 1));
    // symbols that get a fixed share straight away: the tiny ones (low probability) and the ones worth exactly one cell
    bool const isTiny = on && c != 0 && c <= tiny, isOne = on && c > tiny && c <= one;
    n = !on || c == 0 ? 0 : (isTiny ? lowProb : (isOne ? 1 : -2));
    uint32_t placed = (uint32_t)__popcll(__ballot(isTiny || isOne));
    total -= tw_sum((isTiny || isOne) ? c : 0u);
    uint32_t rest = (1u << tableLog) - placed;
    if (rest == 0) return true;
    if (total / rest > one) {                                   // the survivors are rich: raise the "one cell" bar once
        one = (uint32_t)(((uint64_t)total * 3) / ((uint64_t)rest * 2));
        bool const more = n == -2 && c <= one;
        if (more) n = 1;
        placed += (uint32_t)__popcll(__ballot(more));
        total -= tw_sum(more ? c : 0u);
        rest = (1u << tableLog) - placed;
    }
    if (placed == maxSym + 1) {                                 // everybody placed: the most frequent symbol takes what is left
        uint32_t m; uint32_t const big = tw_first_max(c, on, &m);
        if (lane == (big < 64 ? big : 0)) n += (int)rest;
        return true;
    }
    if (total == 0) {                                           // nothing left to weigh: one more cell each, round robin, to the positive ones
        bool const pos = n > 0;
        unsigned long long const pm = __ballot(pos);
        uint32_t const P = (uint32_t)__popcll(pm), r = (uint32_t)__popcll(pm & tw_below(lane));
        if (pos) n += (int)(rest / P + (r < rest % P ? 1u : 0u));
        return true;
    }
    // the unplaced symbols share `rest` cells in proportion to their counts: cell boundaries of a running fixed-point sum
    uint32_t const vLog = 62 - tableLog;
    uint64_t const mid = (1ULL << (vLog - 1)) - 1;
    uint64_t const rStep = ((((uint64_t)1 << vLog) * rest) + mid) / total;
    uint64_t const w = n == -2 ? (uint64_t)c * rStep : 0;
    uint64_t const end = mid + tw_incl64(w), begin = end - w;
    uint32_t const cells = (uint32_t)(end >> vLog) - (uint32_t)(begin >> vLog);
    if (__ballot(n == -2 && cells < 1)) return false;
    if (n == -2) n = (int)cells;
    return true;
}

// FSE_normalizeCount (fse_compress.c:465-525): lane s normalises count[s]; norm[] in LDS.  1 ok, 0 single symbol, -1 error
__device__ inline int fse_normalize_wave(int16_t* norm, uint32_t tableLog, const uint32_t* count, uint32_t total, uint32_t maxSym, bool useLowProb)
{
    uint32_t const lane = tw_lane();
    if (tableLog < 5 || tableLog > 12 || tableLog < fse_min_table_log(total, maxSym)) return -1;
    bool const on = lane <= maxSym;
    uint32_t const c = on ? count[lane] : 0;
    if (__ballot(on && c == total)) return 0;
    int const lowProb = useLowProb ? -1 : 1;
    uint32_t const scale = 62 - tableLog;
    uint64_t const unit = ((uint64_t)1 << 62) / total;                  // one source symbol in table cells, 2^-scale fixed point
    uint32_t const tiny = total >> tableLog;
    bool const isTiny = on && c != 0 && c <= tiny, weighed = on && c > tiny;
    uint32_t share = 0;
    if (weighed) {
        uint64_t const exact = (uint64_t)c * unit;
        share = (uint32_t)(exact >> scale);
        if (share < 8) {                                                // small shares round up past a share-dependent bar (:475-501)
            uint32_t bar;
            switch (share) { case 0: bar = 0; break; case 1: bar = 473195; break; case 2: bar = 504333; break; case 3: bar = 520860; break;
                             case 4: bar = 550000; break; case 5: bar = 700000; break; case 6: bar = 750000; break; default: bar = 830000; break; }
            if (exact - ((uint64_t)share << scale) > ((uint64_t)bar << (scale - 20))) share++;
        }
    }
    int n = isTiny ? lowProb : (int)share;
    int const still = (int)(1u << tableLog) - (int)tw_sum(isTiny ? 1u : share);
    uint32_t topShare; uint32_t top = tw_first_max(share, weighed && share > 0, &topShare);
    if (top >= 64) top = 0;                                             // nobody weighed: the reference's initial `largest = 0`
    int const nTop = (int)__builtin_amdgcn_readlane((uint32_t)n, (int)top);
    if (-still >= (nTop >> 1)) {                                        // the correction would more than halve the top symbol
        if (!fse_spread_rest_wave(n, c, on, tableLog, total, maxSym, lowProb)) return -1;
    } else if (lane == top) n += still;
    if (on) norm[lane] = (int16_t)n;
    __builtin_amdgcn_wave_barrier();
    return 1;
}

// ------------------------------------------------------------------ NCount table description (fse_compress.c:234-327)
// words: 16 zero-initialisable LDS words of scratch; dst: any byte pointer (LDS or global), >= 64 bytes.  returns size, 0 = error
__device__ inline uint32_t fse_write_ncount_wave(uint32_t* words, uint8_t* dst, const int16_t* norm, uint32_t maxSym, uint32_t tableLog)
{
    uint32_t const lane = tw_lane();
    bool const on = lane <= maxSym;
    int const nv = on ? (int)norm[lane] : 0;
    uint32_t const tsz = 1u << tableLog;
    uint32_t const mag = nv < 0 ? 1u : (uint32_t)nv;
    uint32_t const usedIncl = tw_incl(mag);
    int const remBefore = (int)(tsz + 1) - (int)(usedIncl - mag), remAfter = remBefore - (int)mag;
    // zero runs: the first zero of a block is written as a field; the block's other zeros become a run code in front of
    // the next non-zero symbol
    unsigned long long const zeros = __ballot(on && nv == 0);
    bool const prevZero = lane > 0 && ((zeros >> (lane - 1)) & 1);
    bool const live = on && remBefore > 1;                              // the writer stops once the budget is down to 1
    bool const field = live && !(nv == 0 && prevZero);
    if (lane < 16) words[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    uint64_t bits = 0; uint32_t len = 0;
    if (field && nv != 0 && prevZero) {                                 // run code: zeros directly below this lane, minus the block's first
        unsigned long long const below = ~zeros & tw_below(lane);       // non-zero lanes below
        uint32_t const blockStart = below ? 64u - (uint32_t)__clzll((long long)below) : 0u;
        uint32_t const run = lane - blockStart - 1;
        uint32_t const ones = 16 * (run / 24) + 2 * ((run % 24) / 3);
        bits = ((1ULL << ones) - 1) | ((uint64_t)(run % 3) << ones);
        len = ones + 2;
    }
    if (field) {
        uint32_t threshold = tsz;
        if ((uint32_t)remBefore < threshold) threshold = 1u << hb32((uint32_t)remBefore);
        int const maxV = (int)(2 * threshold - 1) - remBefore;
        int v = nv + 1;
        if (v >= (int)threshold) v += maxV;
        uint32_t const flen = hb32(threshold) + (v < maxV ? 0u : 1u);    // log2(threshold) + 1 bits, one less for the small values
        bits |= (uint64_t)(uint32_t)v << len;
        len += flen;
    }
    if (__ballot(live && remAfter < 1)) return 0;
    {   // the budget must end at exactly 1, on the last symbol
        unsigned long long const lm = __ballot(live);
        if (!lm) return 0;
        uint32_t const last = 63u - (uint32_t)__clzll((long long)lm);
        if ((int)__builtin_amdgcn_readlane((uint32_t)remAfter, (int)last) != 1) return 0;
        if ((zeros >> maxSym) & 1) return 0;                            // a table never ends in a zero (maxSym is the last used symbol)
    }
    uint32_t const endBit = 4 + tw_incl(len), pos = endBit - len;
    if (lane == 0) atomicOr(&words[0], tableLog - 5);
    if (len) {
        uint32_t const w = pos >> 5, sh = pos & 31;
        atomicOr(&words[w], (uint32_t)(bits << sh));
        if (sh + len > 32) atomicOr(&words[w + 1], (uint32_t)(bits >> (32 - sh)));
        if (sh + len > 64) atomicOr(&words[w + 2], (uint32_t)(bits >> (64 - sh)));
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t const size = (__builtin_amdgcn_readlane(endBit, 63) + 7) >> 3;
    if (lane < size) dst[lane] = (uint8_t)(words[lane >> 2] >> (8 * (lane & 3)));
    __builtin_amdgcn_wave_barrier();
    return size;
}

// ------------------------------------------------------------------ FSE encoding table (fse_compress.c:68-214)
// cellSym: >= 1 << tableLog bytes of LDS; first: >= 64 u16 of LDS (per-symbol first slot, then running slot)
__device__ inline void fse_build_ctable_wave(FseCTable* ct, const int16_t* norm, uint32_t maxSym, uint32_t tableLog, uint8_t* cellSym, uint16_t* first)
{
    uint32_t const lane = tw_lane();
    uint32_t const tsz = 1u << tableLog, mask = tsz - 1, stride = (tsz >> 1) + (tsz >> 3) + 3;
    bool const on = lane <= maxSym;
    int const nv = on ? (int)norm[lane] : 0;
    bool const low = nv == -1;
    uint32_t const cells = low ? 1u : (nv > 0 ? (uint32_t)nv : 0u), spreadCells = nv > 0 ? (uint32_t)nv : 0u;
    uint32_t const slot0 = tw_incl(cells) - cells;                      // where the symbol's states start in ct->state
    uint32_t const rank0 = tw_incl(spreadCells) - spreadCells;          // its first rank in the spread order
    unsigned long long const lowMask = __ballot(low);
    uint32_t const top = tsz - 1 - (uint32_t)__popcll(lowMask);         // last cell of the spread area
    if (low) cellSym[tsz - 1 - (uint32_t)__popcll(lowMask & tw_below(lane))] = (uint8_t)lane;
    first[lane] = (uint16_t)rank0;                                      // all 64 lanes: the search below needs a monotone array
    if (lane == 0) ct->tableLog = tableLog;
    if (on) {
        uint32_t db; int32_t df; fse_symbol_transform(tableLog, nv, slot0, &db, &df);
        ct->dBits[lane] = db; ct->dFind[lane] = df;
    }
    __builtin_amdgcn_wave_barrier();
    // spread: the j-th visited cell is (j * stride) mod size; cells above `top` are skipped; the r-th cell kept belongs to
    // the last symbol whose first rank is <= r
    uint32_t kept = 0;
    for (uint32_t j0 = 0; j0 < tsz; j0 += 64) {
        uint32_t const j = j0 + lane, cell = (j * stride) & mask;
        bool const ok = j < tsz && cell <= top;
        unsigned long long const okm = __ballot(ok);
        uint32_t const r = kept + (uint32_t)__popcll(okm & tw_below(lane));
        kept += (uint32_t)__popcll(okm);
        if (ok) {
            uint32_t lo = 0, hi = 63;
            while (lo < hi) { uint32_t const mid = (lo + hi + 1) >> 1; if (first[mid] <= r) lo = mid; else hi = mid - 1; }
            cellSym[cell] = (uint8_t)lo;
        }
    }
    __builtin_amdgcn_wave_barrier();
    first[lane] = (uint16_t)slot0;
    __builtin_amdgcn_wave_barrier();
    // states in cell order, per symbol (tableU16[cumul[s]++] = size + u): 64 cells per round, grouped by symbol
    for (uint32_t u0 = 0; u0 < tsz; u0 += 64) {
        uint32_t const u = u0 + lane; bool const inTab = u < tsz;
        uint32_t const sy = inTab ? cellSym[u] : 0xFFu;
        unsigned long long rest = __ballot(inTab);
        while (rest) {
            uint32_t const s = __builtin_amdgcn_readlane(sy, __ffsll((long long)rest) - 1);
            unsigned long long const grp = __ballot(inTab && sy == s);
            uint32_t const base = first[s];
            if (inTab && sy == s) ct->state[base + (uint32_t)__popcll(grp & tw_below(lane))] = (uint16_t)(tsz + u);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) first[s] = (uint16_t)(base + (uint32_t)__popcll(grp));
            __builtin_amdgcn_wave_barrier();
            rest &= ~grp;
        }
    }
    __builtin_amdgcn_wave_barrier();
}
__device__ inline void fse_build_ctable_rle_wave(FseCTable* ct, uint32_t symbol)     // fse_compress.c:528
{
    if (tw_lane() == 0) { ct->tableLog = 0; ct->state[0] = 0; ct->state[1] = 0; ct->dBits[symbol] = 0; ct->dFind[symbol] = 0; }
    __builtin_amdgcn_wave_barrier();
}
#endif  // !ZHIP_TABLES_HOST_ONLY

// host-side builder of the same table (dictionaries: zhip_cdict_host.h): cells in visiting order, then a stable pass per symbol
__host__ inline void fse_build_ctable_host(FseCTable* ct, const int16_t* norm, uint32_t maxSym, uint32_t tableLog)
{
    uint32_t const tsz = 1u << tableLog, mask = tsz - 1, stride = (tsz >> 1) + (tsz >> 3) + 3;
    uint8_t cellSym[4096]; uint32_t slot[257];
    uint32_t used = 0, top = tsz;
    ct->tableLog = tableLog;
    for (uint32_t s = 0; s <= maxSym; s++) {
        int const nv = norm[s];
        uint32_t const cells = nv < 0 ? 1u : (uint32_t)nv;
        slot[s] = used;
        fse_symbol_transform(tableLog, nv, used, &ct->dBits[s], &ct->dFind[s]);
        used += cells;
        if (nv < 0) cellSym[--top] = (uint8_t)s;
    }
    {   uint32_t s = 0, left = 0, j = 0;                            // walk the visiting order once, handing cells to symbols in order
        for (uint32_t placed = 0; placed < top; j++) {
            uint32_t const cell = (j * stride) & mask;
            if (cell >= top) continue;
            while (left == 0) { int const nv = norm[s]; left = nv > 0 ? (uint32_t)nv : 0; if (!left) s++; }
            cellSym[cell] = (uint8_t)s; placed++;
            if (--left == 0) s++;
        }
    }
    for (uint32_t u = 0; u < tsz; u++) ct->state[slot[cellSym[u]]++] = (uint16_t)(tsz + u);
}

// lib/common/fse.h:452-476
__host__ __device__ __forceinline__ uint32_t fse_init_state2(const FseCTable* ct, uint32_t symbol)
{
    uint32_t const nbBitsOut = (ct->dBits[symbol] + (1u << 15)) >> 16;
    uint32_t const v = (nbBitsOut << 16) - ct->dBits[symbol];
    return ct->state[(v >> nbBitsOut) + ct->dFind[symbol]];
}

#if !defined(ZHIP_TABLES_HOST_ONLY)
// ------------------------------------------------------------------ Huffman (huff0)
struct HufNode { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nbBits; };

// wave 0's workspace while it builds the literals code; lives in LDS
struct HufWork {
    // two phases that never overlap share the first 4 112 bytes (round 6: 8.6 -> 5.7 KB; the one-wavefront entropy form is bound by how many records a CU's LDS holds):
    // huf_build_codes_wave works on the tree, huf_write_table_wave (afterwards, from code[] alone) on the weights and their FSE coder
    union {
        HufNode  node[514];           // [0] is the sentinel in front of the leaves (huf_compress.c:683)
        struct {
            uint8_t  weights[256];
            // FSE coder of the weights (tableLog <= 6, 13 symbols)
            uint32_t wCount[16];
            int16_t  wNorm[16];
            uint16_t wFirst[64];
            uint8_t  wSym[64];
            uint32_t wWords[16];
            uint16_t wRec[256];           // per weight: nbBits << 12 | value of its FSE step
            uint32_t wBits[72];           // the weights' bitstream before it is copied behind the NCount bytes
            FseCTable wCt;
        };
    };
    uint16_t key[256];            // order keys (bucket << 8 | 255 - symbol), then scratch
    uint32_t bucketCount[192];
    uint8_t  stack[256];          // explicit quicksort stack (pairs)
    uint32_t perLen[16];          // symbols per code length, then the running canonical value per length
    uint32_t bcast[2];            // the builder's two wave-wide scalars (height limit result, last used symbol)
};
static_assert(sizeof(HufNode) * 514 >= 256 + 64 + 32 + 128 + 64 + 64 + 512 + 288 + sizeof(FseCTable), "the weights' phase fits under the tree");


__device__ __forceinline__ uint32_t huf_bucket(uint32_t c) { return c < 165 ? c : hb32(c) + 158; }   // huf_compress.c:530 (HUF_getIndex)

// the reference's in-bucket order for counts >= 166 is whatever its quicksort leaves (unstable among equal counts): same
// partition scheme, same pivot, same small-range insertion sort, on one lane (huf_compress.c:555-607)
__device__ inline void huf_bucket_sort_serial(HufNode* a, int lo0, int hi0, uint8_t* stack)
{
    int sp = 0;
    stack[sp++] = (uint8_t)lo0; stack[sp++] = (uint8_t)hi0;
    while (sp) {
        int hi = stack[--sp], lo = stack[--sp];
        if (hi - lo < 8) {                                      // insertion sort, decreasing
            for (int i = lo + 1; i <= hi; i++) {
                HufNode const k = a[i]; int j = i - 1;
                while (j >= lo && a[j].count < k.count) { a[j + 1] = a[j]; j--; }
                a[j + 1] = k;
            }
            continue;
        }
        while (lo < hi) {
            uint32_t const pivot = a[hi].count; int i = lo - 1;
            for (int j = lo; j < hi; j++) if (a[j].count > pivot) { i++; HufNode const t = a[i]; a[i] = a[j]; a[j] = t; }
            {   HufNode const t = a[i + 1]; a[i + 1] = a[hi]; a[hi] = t; }
            int const p = i + 1;
            if (p - lo < hi - p) { if (p - 1 > lo) { stack[sp++] = (uint8_t)lo; stack[sp++] = (uint8_t)(p - 1); } lo = p + 1; }
            else { if (hi > p + 1) { stack[sp++] = (uint8_t)(p + 1); stack[sp++] = (uint8_t)hi; } hi = p - 1; }
        }
    }
}

// HUF_setMaxHeight (huf_compress.c:376-498) on one lane: which symbols pay for the cut is defined by the reference's own walk
__device__ inline uint32_t huf_limit_height_serial(HufNode* node, uint32_t lastNonNull, uint32_t target)
{
    uint32_t const deepest = node[lastNonNull].nbBits;
    if (deepest <= target) return deepest;
    uint32_t const none = 0xF0F0F0F0u;
    int debt = 0, n = (int)lastNonNull;
    for (; node[n].nbBits > target; n--) { debt += (int)((1u << (deepest - target)) - (1u << (deepest - node[n].nbBits))); node[n].nbBits = (uint8_t)target; }
    while (node[n].nbBits == target) n--;
    debt >>= (deepest - target);
    uint32_t lastOf[14];                                       // lastOf[d]: last leaf whose length is target - d
    for (int i = 0; i < 14; i++) lastOf[i] = none;
    for (int pos = n, cur = (int)target; pos >= 0; pos--) if (node[pos].nbBits < (uint32_t)cur) { cur = node[pos].nbBits; lastOf[target - (uint32_t)cur] = (uint32_t)pos; }
    while (debt > 0) {
        uint32_t d = hb32((uint32_t)debt) + 1;
        for (; d > 1; d--) {
            uint32_t const hi = lastOf[d], lo = lastOf[d - 1];
            if (hi == none) continue;
            if (lo == none || node[hi].count <= 2 * node[lo].count) break;
        }
        while (d <= 12 && lastOf[d] == none) d++;
        debt -= 1 << (d - 1);
        node[lastOf[d]].nbBits++;
        if (lastOf[d - 1] == none) lastOf[d - 1] = lastOf[d];
        if (lastOf[d] == 0) lastOf[d] = none;
        else if (node[--lastOf[d]].nbBits != target - d) lastOf[d] = none;
    }
    for (; debt < 0; debt++) {
        if (lastOf[1] == none) { while (node[n].nbBits == target) n--; node[n + 1].nbBits--; lastOf[1] = (uint32_t)(n + 1); }
        else { node[lastOf[1] + 1].nbBits--; lastOf[1]++; }
    }
    return target;
}

// HUF_buildCTable_wksp (huf_compress.c:756-791).  code[s] = value << 8 | nbBits.  returns the table log actually used.
__device__ inline uint32_t huf_build_codes_wave(HufWork* w, const uint32_t* count, uint32_t maxSym, uint32_t maxNbBits, uint32_t* code)
{
    uint32_t const lane = tw_lane();
    HufNode* const node0 = w->node; HufNode* const node = w->node + 1;
    {   uint64_t* const z = (uint64_t*)node0;                   // 514 nodes of 8 bytes
        for (uint32_t i = lane; i < 514; i += 64) z[i] = 0;
        for (uint32_t i = lane; i < 192; i += 64) w->bucketCount[i] = 0;
        if (lane < 16) w->perLen[lane] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    // 1. order by decreasing count the reference's way (HUF_sort, :620-665): by bucket, inside a bucket by symbol — the
    //    position of a symbol is the number of symbols whose key is larger
    for (uint32_t s = lane; s <= maxSym; s += 64) {
        uint32_t const b = huf_bucket(count[s]);
        w->key[s] = (uint16_t)((b << 8) | (255 - s));
        atomicAdd(&w->bucketCount[b], 1u);
    }
    __builtin_amdgcn_wave_barrier();
    {   uint32_t kq[4], above[4] = { 0, 0, 0, 0 };
        for (int q = 0; q < 4; q++) { uint32_t const s = lane + 64u * (uint32_t)q; kq[q] = s <= maxSym ? w->key[s] : 0xFFFFu; }
        for (uint32_t m = 0; m <= maxSym; m++) {
            uint32_t const km = w->key[m];                      // broadcast read
            for (int q = 0; q < 4; q++) above[q] += km > kq[q] ? 1u : 0u;
        }
        for (int q = 0; q < 4; q++) {
            uint32_t const s = lane + 64u * (uint32_t)q;
            if (s <= maxSym) { node[above[q]].count = count[s]; node[above[q]].byte = (uint8_t)s; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        // buckets 164 .. 189: the reference sorts rankPosition[CUTOFF .. 190] (:655-664), CUTOFF = 158 + highbit32(158) = 165
        // (its comment says 166), and rankPosition[n] holds the symbols of bucket n - 1 — so the symbols whose count is
        // exactly 164 go through the (unstable) quicksort too.  Their members, now in symbol order, get that order
        uint32_t start = 0;
        for (int b = 191; b >= 164; b--) {
            uint32_t const sz = w->bucketCount[b];
            if (b <= 189 && sz > 1) huf_bucket_sort_serial(node + start, 0, (int)sz - 1, w->stack);
            start += sz;
        }
        // 2. the two-queue merge (HUF_buildTree, :681-718): leaves from the rare end, internal nodes in creation order
        int last = (int)maxSym;
        while (node[last].count == 0) last--;
        int leaf = last, made = 256, take = 256; int const root = 256 + last - 1;
        node[made].count = node[leaf].count + node[leaf - 1].count;
        node[leaf].parent = node[leaf - 1].parent = (uint16_t)made;
        made++; leaf -= 2;
        for (int i = made; i <= root; i++) node[i].count = 1u << 30;
        node0[0].count = 1u << 31;
        while (made <= root) {
            int const a = node[leaf].count < node[take].count ? leaf-- : take++;
            int const b = node[leaf].count < node[take].count ? leaf-- : take++;
            node[made].count = node[a].count + node[b].count;
            node[a].parent = node[b].parent = (uint16_t)made;
            made++;
        }
        w->bcast[1] = (uint32_t)last;
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t const last = w->bcast[1], root = 256 + last - 1;
    // 3. depths from the parent links: internal nodes from the root downwards, 64 at a time (a parent has a higher index;
    //    one inside the same group is caught by repeating the group until nothing changes), then the leaves
    for (int hiN = (int)root - 1; hiN >= 256; hiN -= 64) {
        int const nIdx = hiN - (int)lane; bool const in = nIdx >= 256;
        for (;;) {
            uint32_t want = 0, have = 0;
            if (in) { want = node[node[nIdx].parent].nbBits + 1u; have = node[nIdx].nbBits; }
            __builtin_amdgcn_wave_barrier();
            if (in && want != have) node[nIdx].nbBits = (uint8_t)want;
            __builtin_amdgcn_wave_barrier();
            if (!__ballot(in && want != have)) break;
        }
    }
    for (uint32_t n = lane; n <= last; n += 64) node[n].nbBits = (uint8_t)(node[node[n].parent].nbBits + 1);
    __builtin_amdgcn_wave_barrier();
    // 4. height limit
    if (lane == 0) w->bcast[0] = huf_limit_height_serial(node, last, maxNbBits);
    __builtin_amdgcn_wave_barrier();
    maxNbBits = w->bcast[0];
    // 5. canonical values (HUF_buildCTableFromTree, :730-753): per length, values count up in symbol order
    for (uint32_t n = lane; n <= last; n += 64) atomicAdd(&w->perLen[node[n].nbBits], 1u);
    for (uint32_t s = lane; s < 256; s += 64) code[s] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t n = lane; n <= maxSym; n += 64) code[node[n].byte] = node[n].nbBits;
    if (lane == 0) {
        uint32_t startVal = 0;
        for (uint32_t len = maxNbBits; len > 0; len--) { uint32_t const cnt = w->perLen[len]; w->perLen[len] = startVal; startVal = (startVal + cnt) >> 1; }
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t s0 = 0; s0 <= maxSym; s0 += 64) {
        uint32_t const s = s0 + lane; bool const in = s <= maxSym;
        uint32_t const len = in ? code[s] : 0;
        unsigned long long rest = __ballot(in && len != 0);
        while (rest) {
            uint32_t const L = __builtin_amdgcn_readlane(len, __ffsll((long long)rest) - 1);
            unsigned long long const grp = __ballot(in && len == L);
            uint32_t const base = w->perLen[L];
            if (in && len == L) code[s] = ((base + (uint32_t)__popcll(grp & tw_below(lane))) << 8) | L;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) w->perLen[L] = base + (uint32_t)__popcll(grp);
            __builtin_amdgcn_wave_barrier();
            rest &= ~grp;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return maxNbBits;
}

// one FSE encoding step: (nbBits << 12 | low bits of the state) and the next state
__device__ __forceinline__ uint32_t fse_step_record(const FseCTable* ct, uint32_t& state, uint32_t symbol)
{
    uint32_t const nb = (state + ct->dBits[symbol]) >> 16;
    uint32_t const rec = (nb << 12) | (state & ((1u << nb) - 1));
    state = ct->state[(state >> nb) + ct->dFind[symbol]];
    return rec;
}

// HUF_compressWeights (huf_compress.c:147-186): NCount + the weights coded backwards with two alternating FSE states
// (fse_compress.c:551-608).  dst: byte pointer (LDS).  0 = not compressible
__device__ inline uint32_t huf_compress_weights_wave(HufWork* w, uint8_t* dst, const uint8_t* wt, uint32_t n)
{
    uint32_t const lane = tw_lane();
    if (n <= 1) return 0;
    if (n == 2) return wt[0] == wt[1];
    if (lane < 16) w->wCount[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < n; i += 64) atomicAdd(&w->wCount[wt[i]], 1u);
    __builtin_amdgcn_wave_barrier();
    uint32_t const c = lane <= 12 ? w->wCount[lane] : 0;
    unsigned long long const used = __ballot(c != 0);
    uint32_t const maxSym = 63u - (uint32_t)__clzll((long long)used), maxCount = tw_max(c);
    if (maxCount == n) return 1;
    if (maxCount == 1) return 0;
    uint32_t const tableLog = fse_optimal_table_log(6, n, maxSym, 2);
    if (fse_normalize_wave(w->wNorm, tableLog, w->wCount, n, maxSym, false) < 0) return 0;
    uint32_t const hdr = fse_write_ncount_wave(w->wWords, dst, w->wNorm, maxSym, tableLog);
    if (!hdr) return 0;
    fse_build_ctable_wave(&w->wCt, w->wNorm, maxSym, tableLog, w->wSym, w->wFirst);
    // backwards index t (0 = last weight): t = 0, 1 start the two states, every later t is one step of the state with t's
    // parity (the even-t state is "state 1" when n is odd, "state 2" when n is even); lanes 0 / 1 walk the two chains
    const FseCTable* const ct = &w->wCt;
    uint32_t fin = 0;
    if (lane < 2) {
        uint32_t st = fse_init_state2(ct, wt[n - 1 - lane]);
        for (uint32_t t = lane + 2; t < n; t += 2) w->wRec[t] = (uint16_t)fse_step_record(ct, st, wt[n - 1 - t]);
        fin = st;
    }
    __builtin_amdgcn_wave_barrier();
    // bit layout: steps in t order, then the two final states (state 2 first), then the end mark
    uint32_t const fin0 = __builtin_amdgcn_readlane(fin, 0), fin1 = __builtin_amdgcn_readlane(fin, 1);
    uint32_t const finA = (n & 1) ? fin1 : fin0, finB = (n & 1) ? fin0 : fin1;          // state 2, state 1
    uint32_t const steps = n - 2;
    for (uint32_t i = lane; i < 72; i += 64) w->wBits[i] = 0;
    __builtin_amdgcn_wave_barrier();
    uint32_t base = 0;
    for (uint32_t t0 = 0; t0 < steps + 3; t0 += 64) {
        uint32_t const k = t0 + lane;
        uint32_t v = 0, len = 0;
        if (k < steps) { uint32_t const r = w->wRec[k + 2]; len = r >> 12; v = r & 0xFFFu; }
        else if (k == steps) { v = finA & ((1u << tableLog) - 1); len = tableLog; }          // FSE_flushCState: the low tableLog bits
        else if (k == steps + 1) { v = finB & ((1u << tableLog) - 1); len = tableLog; }
        else if (k == steps + 2) { v = 1; len = 1; }
        uint32_t const endPos = base + tw_incl(len), pos = endPos - len;
        if (len) {
            atomicOr(&w->wBits[pos >> 5], v << (pos & 31));
            if ((pos & 31) + len > 32) atomicOr(&w->wBits[(pos >> 5) + 1], v >> (32 - (pos & 31)));
        }
        base = __builtin_amdgcn_readlane(endPos, 63);
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t const bytes = (base + 7) >> 3;
    for (uint32_t i = lane; i < bytes; i += 64) dst[hdr + i] = (uint8_t)(w->wBits[i >> 2] >> (8 * (i & 3)));
    __builtin_amdgcn_wave_barrier();
    return hdr + bytes;
}

// HUF_writeCTable_wksp (huf_compress.c:248-289). dst: >= 136 bytes of LDS. returns size, 0 on failure
__device__ inline uint32_t huf_write_table_wave(HufWork* w, uint8_t* dst, const uint32_t* code, uint32_t maxSym, uint32_t huffLog)
{
    uint32_t const lane = tw_lane();
    for (uint32_t s = lane; s < maxSym; s += 64) { uint32_t const len = code[s] & 0xFF; w->weights[s] = len ? (uint8_t)(huffLog + 1 - len) : 0; }
    if (lane == 0) w->weights[maxSym] = 0;
    __builtin_amdgcn_wave_barrier();
    uint32_t const h = huf_compress_weights_wave(w, dst + 1, w->weights, maxSym);
    if (h > 1 && h < maxSym / 2) { if (lane == 0) dst[0] = (uint8_t)h; __builtin_amdgcn_wave_barrier(); return h + 1; }
    if (maxSym > 128) return 0;
    if (lane == 0) dst[0] = (uint8_t)(128 + (maxSym - 1));
    for (uint32_t p = lane; 2 * p < maxSym; p += 64) dst[p + 1] = (uint8_t)((w->weights[2 * p] << 4) + w->weights[2 * p + 1]);
    __builtin_amdgcn_wave_barrier();
    return ((maxSym + 1) / 2) + 1;
}
#endif  // !ZHIP_TABLES_HOST_ONLY

}  // namespace zhip
