// zhip_tables.h — entropy *table construction* for the gfx950 entropy kernel: Huffman code lengths (huff0),
// FSE normalisation / NCount header / encoding tables.  These are small (<= 256 symbols), branchy, strictly
// ordered computations, so each is executed by ONE lane working on LDS-resident arrays while the rest of the
// workgroup waits or builds another table; the data-parallel work (histograms, bit packing) is in zhip_entropy.h.
//
// Every decision follows the reference bit for bit (cited per function; paths relative to facebook/zstd).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zhip {

__host__ __device__ __forceinline__ uint32_t hb32(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }   // lib/common/bits.h:177

// ------------------------------------------------------------------ FSE
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

// lib/compress/fse_compress.c:348-369
__device__ inline uint32_t fse_min_table_log(uint32_t n, uint32_t maxSym)
{
    uint32_t const a = hb32(n) + 1, b = hb32(maxSym) + 2;
    return a < b ? a : b;
}
__device__ inline uint32_t fse_optimal_table_log(uint32_t maxLog, uint32_t n, uint32_t maxSym, uint32_t minus)
{
    uint32_t const maxBitsSrc = hb32(n - 1) - minus;
    uint32_t log = maxLog, minBits = fse_min_table_log(n, maxSym);
    if (maxBitsSrc < log) log = maxBitsSrc;
    if (minBits > log) log = minBits;
    if (log < 5) log = 5;
    if (log > 12) log = 12;
    return log;
}

// lib/compress/fse_compress.c:379-463 (secondary normalisation). returns false on failure
__device__ inline bool fse_normalize_m2(int16_t* norm, uint32_t tableLog, const uint32_t* count, uint32_t total,
                                        uint32_t maxSym, int16_t lowProb)
{
    uint32_t s, distributed = 0, toDistribute;
    uint32_t const lowThreshold = total >> tableLog;
    uint32_t lowOne = (uint32_t)(((uint64_t)total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = -2;
    }
    toDistribute = (1u << tableLog) - distributed;
    if (toDistribute == 0) return true;
    if ((total / toDistribute) > lowOne) {
        lowOne = (uint32_t)(((uint64_t)total * 3) / (toDistribute * 2));
        for (s = 0; s <= maxSym; s++)
            if (norm[s] == -2 && count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; }
        toDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSym + 1) {
        uint32_t maxV = 0, maxC = 0;
        for (s = 0; s <= maxSym; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] = (int16_t)(norm[maxV] + (int16_t)toDistribute);
        return true;
    }
    if (total == 0) {
        for (s = 0; toDistribute > 0; s = (s + 1) % (maxSym + 1))
            if (norm[s] > 0) { toDistribute--; norm[s]++; }
        return true;
    }
    {   uint64_t const vStepLog = 62 - tableLog;
        uint64_t const mid = (1ULL << (vStepLog - 1)) - 1;
        uint64_t const rStep = ((((uint64_t)1 << vStepLog) * toDistribute) + mid) / total;
        uint64_t tmpTotal = mid;
        for (s = 0; s <= maxSym; s++) {
            if (norm[s] == -2) {
                uint64_t const end = tmpTotal + ((uint64_t)count[s] * rStep);
                uint32_t const sStart = (uint32_t)(tmpTotal >> vStepLog), sEnd = (uint32_t)(end >> vStepLog);
                if (sEnd - sStart < 1) return false;
                norm[s] = (int16_t)(sEnd - sStart);
                tmpTotal = end;
    }   }   }
    return true;
}

// lib/compress/fse_compress.c:465-525. returns 1 ok, 0 rle special case, -1 error
__device__ inline int fse_normalize(int16_t* norm, uint32_t tableLog, const uint32_t* count, uint32_t total,
                                    uint32_t maxSym, bool useLowProb)
{
    const uint32_t rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    int16_t const lowProb = useLowProb ? -1 : 1;
    uint64_t const scale = 62 - tableLog;
    uint64_t const step = ((uint64_t)1 << 62) / total;
    uint64_t const vStep = 1ULL << (scale - 20);
    int still = 1 << tableLog;
    uint32_t s, largest = 0; int16_t largestP = 0;
    uint32_t const lowThreshold = total >> tableLog;
    if (tableLog < 5 || tableLog > 12) return -1;
    if (tableLog < fse_min_table_log(total, maxSym)) return -1;
    for (s = 0; s <= maxSym; s++) {
        if (count[s] == total) return 0;
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProb; still--; }
        else {
            int16_t proba = (int16_t)(((uint64_t)count[s] * step) >> scale);
            if (proba < 8) {
                uint64_t const restToBeat = vStep * rtb[proba];
                proba = (int16_t)(proba + ((((uint64_t)count[s] * step) - ((uint64_t)proba << scale)) > restToBeat));
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba; still -= proba;
        }
    }
    if (-still >= (norm[largest] >> 1)) {
        if (!fse_normalize_m2(norm, tableLog, count, total, maxSym, lowProb)) return -1;
    } else norm[largest] = (int16_t)(norm[largest] + (int16_t)still);
    return 1;
}

// lib/compress/fse_compress.c:234-327 (write-is-safe path; caller provides >= 64 bytes). returns size, 0 on error
__device__ inline uint32_t fse_write_ncount(uint8_t* out0, const int16_t* norm, uint32_t maxSym, uint32_t tableLog)
{
    uint8_t* out = out0;
    int const tableSize = 1 << tableLog;
    int nbBits = (int)tableLog + 1, remaining = tableSize + 1, threshold = tableSize, bitCount = 4;
    bool previousIs0 = false;
    uint32_t bitStream = tableLog - 5, symbol = 0;
    uint32_t const alphabetSize = maxSym + 1;
    while (symbol < alphabetSize && remaining > 1) {
        if (previousIs0) {
            uint32_t start = symbol;
            while (symbol < alphabetSize && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24; bitStream += 0xFFFFU << bitCount;
                out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16;
            }
            while (symbol >= start + 3) { start += 3; bitStream += 3U << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) { out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
        }
        {   int count = norm[symbol++];
            int const max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (uint32_t)count << bitCount;
            bitCount += nbBits; bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) { out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
    }
    if (remaining != 1) return 0;
    out[0] = (uint8_t)bitStream; out[1] = (uint8_t)(bitStream >> 8);
    out += (bitCount + 7) / 8;
    return (uint32_t)(out - out0);
}

// lib/compress/fse_compress.c:68-214. symScratch: >= 1<<tableLog bytes, cumul: >= maxSym+2 u16
__host__ __device__ inline void fse_build_ctable(FseCTable* ct, const int16_t* norm, uint32_t maxSym, uint32_t tableLog,
                                        uint8_t* symScratch, uint16_t* cumul)
{
    uint32_t const tableSize = 1u << tableLog, mask = tableSize - 1;
    uint32_t const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    uint32_t high = tableSize - 1, u, pos = 0, s, total = 0;
    ct->tableLog = tableLog;
    cumul[0] = 0;
    for (u = 1; u <= maxSym + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = (uint16_t)(cumul[u - 1] + 1); symScratch[high--] = (uint8_t)(u - 1); }
        else cumul[u] = (uint16_t)(cumul[u - 1] + (uint16_t)norm[u - 1]);
    }
    for (s = 0; s <= maxSym; s++) {
        for (int i = 0; i < norm[s]; i++) {
            symScratch[pos] = (uint8_t)s;
            pos = (pos + step) & mask;
            while (pos > high) pos = (pos + step) & mask;
        }
    }
    for (u = 0; u < tableSize; u++) { uint8_t const c = symScratch[u]; ct->state[cumul[c]++] = (uint16_t)(tableSize + u); }
    for (s = 0; s <= maxSym; s++) {
        int const nv = norm[s];
        if (nv == 0) { ct->dBits[s] = ((tableLog + 1) << 16) - (1u << tableLog); ct->dFind[s] = 0; }
        else if (nv == -1 || nv == 1) { ct->dBits[s] = (tableLog << 16) - (1u << tableLog); ct->dFind[s] = (int32_t)(total - 1); total++; }
        else {
            uint32_t const maxBitsOut = tableLog - hb32((uint32_t)nv - 1);
            uint32_t const minStatePlus = (uint32_t)nv << maxBitsOut;
            ct->dBits[s] = (maxBitsOut << 16) - minStatePlus;
            ct->dFind[s] = (int32_t)(total - (uint32_t)nv);
            total += (uint32_t)nv;
        }
    }
}
__device__ inline void fse_build_ctable_rle(FseCTable* ct, uint32_t symbol)          // fse_compress.c:528
{
    ct->tableLog = 0; ct->state[0] = 0; ct->state[1] = 0; ct->dBits[symbol] = 0; ct->dFind[symbol] = 0;
}
// lib/common/fse.h:452-476
__device__ __forceinline__ uint32_t fse_init_state2(const FseCTable* ct, uint32_t symbol)
{
    uint32_t const nbBitsOut = (ct->dBits[symbol] + (1u << 15)) >> 16;
    uint32_t const v = (nbBitsOut << 16) - ct->dBits[symbol];
    return ct->state[(v >> nbBitsOut) + ct->dFind[symbol]];
}

// minimal LSB-first byte writer for the tiny single-lane streams (table headers)
struct BitW { uint8_t* p; uint64_t acc; uint32_t nb; };
__device__ __forceinline__ void bw_add(BitW& b, uint64_t v, uint32_t n)
{
    if (n == 0) return;
    b.acc |= (v & ((1ULL << n) - 1)) << b.nb; b.nb += n;
    while (b.nb >= 8) { *b.p++ = (uint8_t)b.acc; b.acc >>= 8; b.nb -= 8; }
}
__device__ __forceinline__ uint8_t* bw_close(BitW& b)        // lib/common/bitstream.h:222
{
    bw_add(b, 1, 1);
    if (b.nb) { *b.p++ = (uint8_t)b.acc; b.acc = 0; b.nb = 0; }
    return b.p;
}
__device__ __forceinline__ uint32_t fse_encode_sym(BitW& b, const FseCTable* ct, uint32_t state, uint32_t symbol)
{
    uint32_t const nbBitsOut = (state + ct->dBits[symbol]) >> 16;
    bw_add(b, state, nbBitsOut);
    return ct->state[(state >> nbBitsOut) + ct->dFind[symbol]];
}

// ------------------------------------------------------------------ Huffman (huff0)
struct HufNode { uint32_t count; uint16_t parent; uint8_t byte; uint8_t nbBits; };
struct HufRank { uint16_t base, curr; };

// workspace of the single lane that builds the literals code; lives in LDS
struct HufWork {
    HufNode  node[514];           // [0] is the sentinel in front of huffNode (huf_compress.c:683)
    HufRank  rank[192];
    uint8_t  stack[256];          // explicit quicksort stack (pairs)
    uint8_t  weights[256];
    // FSE coder of the weights (tableLog <= 6, 13 symbols)
    uint32_t wCount[16];
    int16_t  wNorm[16];
    uint16_t wCumul[18];
    uint8_t  wSym[64];
    FseCTable wCt;
};

__device__ __forceinline__ uint32_t huf_bucket(uint32_t c) { return c < 166 ? c : hb32(c) + 158; }   // huf_compress.c:530

__device__ inline void huf_isort(HufNode* a, int low, int high)                    // huf_compress.c:555
{
    int const size = high - low + 1; a += low;
    for (int i = 1; i < size; i++) {
        HufNode const key = a[i]; int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}
__device__ inline int huf_partition(HufNode* a, int low, int high)                 // huf_compress.c:571
{
    uint32_t const pivot = a[high].count; int i = low - 1; HufNode t;
    for (int j = low; j < high; j++) if (a[j].count > pivot) { i++; t = a[i]; a[i] = a[j]; a[j] = t; }
    t = a[i + 1]; a[i + 1] = a[high]; a[high] = t;
    return i + 1;
}
// huf_compress.c:591-607, recursion replaced by an explicit stack: sub-ranges are disjoint, so the order in which
// they are finished cannot change the result.  The entry test (insertion sort below 8) applies to *calls* only.
__device__ inline void huf_qsort(HufNode* a, int low0, int high0, uint8_t* stack)
{
    int sp = 0;
    stack[sp++] = (uint8_t)low0; stack[sp++] = (uint8_t)high0;
    while (sp) {
        int high = stack[--sp], low = stack[--sp];
        if (high - low < 8) { huf_isort(a, low, high); continue; }
        while (low < high) {
            int const idx = huf_partition(a, low, high);
            if (idx - low < high - idx) { if (idx - 1 > low) { stack[sp++] = (uint8_t)low; stack[sp++] = (uint8_t)(idx - 1); } low = idx + 1; }
            else { if (high > idx + 1) { stack[sp++] = (uint8_t)(idx + 1); stack[sp++] = (uint8_t)high; } high = idx - 1; }
        }
    }
}

// huf_compress.c:376-498
__device__ inline uint32_t huf_set_max_height(HufNode* node, uint32_t lastNonNull, uint32_t target)
{
    uint32_t const largestBits = node[lastNonNull].nbBits;
    if (largestBits <= target) return largestBits;
    int totalCost = 0, n = (int)lastNonNull;
    uint32_t const baseCost = 1u << (largestBits - target);
    uint32_t const noSymbol = 0xF0F0F0F0u;
    uint32_t rankLast[14];
    while (node[n].nbBits > target) {
        totalCost += (int)(baseCost - (1u << (largestBits - node[n].nbBits)));
        node[n].nbBits = (uint8_t)target; n--;
    }
    while (node[n].nbBits == target) --n;
    totalCost >>= (largestBits - target);
    for (int i = 0; i < 14; i++) rankLast[i] = noSymbol;
    {   uint32_t cur = target;
        for (int pos = n; pos >= 0; pos--) {
            if (node[pos].nbBits >= cur) continue;
            cur = node[pos].nbBits;
            rankLast[target - cur] = (uint32_t)pos;
    }   }
    while (totalCost > 0) {
        uint32_t nBitsToDecrease = hb32((uint32_t)totalCost) + 1;
        for ( ; nBitsToDecrease > 1; nBitsToDecrease--) {
            uint32_t const highPos = rankLast[nBitsToDecrease], lowPos = rankLast[nBitsToDecrease - 1];
            if (highPos == noSymbol) continue;
            if (lowPos == noSymbol) break;
            if (node[highPos].count <= 2 * node[lowPos].count) break;
        }
        while (nBitsToDecrease <= 12 && rankLast[nBitsToDecrease] == noSymbol) nBitsToDecrease++;
        totalCost -= 1 << (nBitsToDecrease - 1);
        node[rankLast[nBitsToDecrease]].nbBits++;
        if (rankLast[nBitsToDecrease - 1] == noSymbol) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
        if (rankLast[nBitsToDecrease] == 0) rankLast[nBitsToDecrease] = noSymbol;
        else {
            rankLast[nBitsToDecrease]--;
            if (node[rankLast[nBitsToDecrease]].nbBits != target - nBitsToDecrease) rankLast[nBitsToDecrease] = noSymbol;
        }
    }
    while (totalCost < 0) {
        if (rankLast[1] == noSymbol) {
            while (node[n].nbBits == target) n--;
            node[n + 1].nbBits--; rankLast[1] = (uint32_t)(n + 1); totalCost++;
            continue;
        }
        node[rankLast[1] + 1].nbBits--; rankLast[1]++; totalCost++;
    }
    return target;
}

// huf_compress.c:756-791 = sort (:620-665) + tree (:681-718) + height limit + canonical codes (:730-753).
// code[s] = value << 8 | nbBits.  returns the table log actually used.
__device__ inline uint32_t huf_build_codes(HufWork* w, const uint32_t* count, uint32_t maxSym, uint32_t maxNbBits, uint32_t* code)
{
    HufNode* const node0 = w->node; HufNode* const node = w->node + 1;
    for (int i = 0; i < 514; i++) { node0[i].count = 0; node0[i].parent = 0; node0[i].byte = 0; node0[i].nbBits = 0; }
    // sort by decreasing count, reference bucket order
    for (int i = 0; i < 192; i++) { w->rank[i].base = 0; w->rank[i].curr = 0; }
    for (uint32_t n = 0; n <= maxSym; n++) w->rank[huf_bucket(count[n])].base++;
    for (int n = 191; n > 0; n--) { w->rank[n - 1].base = (uint16_t)(w->rank[n - 1].base + w->rank[n].base); w->rank[n - 1].curr = w->rank[n - 1].base; }
    for (uint32_t n = 0; n <= maxSym; n++) {
        uint32_t const r = huf_bucket(count[n]) + 1;
        uint32_t const pos = w->rank[r].curr++;
        node[pos].count = count[n]; node[pos].byte = (uint8_t)n;
    }
    for (int n = 166; n < 191; n++) {
        int const sz = (int)w->rank[n].curr - (int)w->rank[n].base;
        if (sz > 1) huf_qsort(node + w->rank[n].base, 0, sz - 1, w->stack);
    }
    // tree
    int nonNull = (int)maxSym, lowS, lowN, nodeNb = 256, nodeRoot, n;
    while (node[nonNull].count == 0) nonNull--;
    lowS = nonNull; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
    node[nodeNb].count = node[lowS].count + node[lowS - 1].count;
    node[lowS].parent = node[lowS - 1].parent = (uint16_t)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) node[n].count = 1u << 30;
    node0[0].count = 1u << 31;
    while (nodeNb <= nodeRoot) {
        int const n1 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        int const n2 = (node[lowS].count < node[lowN].count) ? lowS-- : lowN++;
        node[nodeNb].count = node[n1].count + node[n2].count;
        node[n1].parent = node[n2].parent = (uint16_t)nodeNb;
        nodeNb++;
    }
    node[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= 256; n--) node[n].nbBits = (uint8_t)(node[node[n].parent].nbBits + 1);
    for (n = 0; n <= nonNull; n++) node[n].nbBits = (uint8_t)(node[node[n].parent].nbBits + 1);
    maxNbBits = huf_set_max_height(node, (uint32_t)nonNull, maxNbBits);
    // canonical values
    uint16_t nbPerRank[13], valPerRank[13]; uint16_t mn = 0;
    for (n = 0; n < 13; n++) { nbPerRank[n] = 0; valPerRank[n] = 0; }
    for (n = 0; n <= nonNull; n++) nbPerRank[node[n].nbBits]++;
    for (n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = mn; mn = (uint16_t)(mn + nbPerRank[n]); mn >>= 1; }
    for (n = 0; n < 256; n++) code[n] = 0;
    for (n = 0; n <= (int)maxSym; n++) code[node[n].byte] = node[n].nbBits;
    for (n = 0; n <= (int)maxSym; n++) { uint32_t const nb = code[n]; if (nb) code[n] = ((uint32_t)valPerRank[nb]++ << 8) | nb; }
    return maxNbBits;
}

// huf_compress.c:147-186 HUF_compressWeights (+ fse_compress.c:551-608, two interleaved states). 0 = not compressible
__device__ inline uint32_t huf_compress_weights(HufWork* w, uint8_t* dst, const uint8_t* wt, uint32_t n)
{
    if (n <= 1) return 0;
    if (n == 2) return wt[0] == wt[1];
    uint32_t maxSym = 12, maxCount = 0;
    for (uint32_t s = 0; s <= 12; s++) w->wCount[s] = 0;
    for (uint32_t i = 0; i < n; i++) w->wCount[wt[i]]++;
    while (!w->wCount[maxSym]) maxSym--;
    for (uint32_t s = 0; s <= maxSym; s++) if (w->wCount[s] > maxCount) maxCount = w->wCount[s];
    if (maxCount == n) return 1;
    if (maxCount == 1) return 0;
    uint32_t const tableLog = fse_optimal_table_log(6, n, maxSym, 2);
    if (fse_normalize(w->wNorm, tableLog, w->wCount, n, maxSym, false) < 0) return 0;
    uint8_t* op = dst;
    {   uint32_t const h = fse_write_ncount(op, w->wNorm, maxSym, tableLog);
        if (!h) return 0;
        op += h;
    }
    fse_build_ctable(&w->wCt, w->wNorm, maxSym, tableLog, w->wSym, w->wCumul);
    BitW b; b.p = op; b.acc = 0; b.nb = 0;
    uint32_t i = n, s1, s2;
    if (n & 1) { s1 = fse_init_state2(&w->wCt, wt[i - 1]); s2 = fse_init_state2(&w->wCt, wt[i - 2]); s1 = fse_encode_sym(b, &w->wCt, s1, wt[i - 3]); i -= 3; }
    else       { s2 = fse_init_state2(&w->wCt, wt[i - 1]); s1 = fse_init_state2(&w->wCt, wt[i - 2]); i -= 2; }
    if ((n - 2) & 2) { s2 = fse_encode_sym(b, &w->wCt, s2, wt[i - 1]); s1 = fse_encode_sym(b, &w->wCt, s1, wt[i - 2]); i -= 2; }
    while (i >= 4) {
        s2 = fse_encode_sym(b, &w->wCt, s2, wt[i - 1]); s1 = fse_encode_sym(b, &w->wCt, s1, wt[i - 2]);
        s2 = fse_encode_sym(b, &w->wCt, s2, wt[i - 3]); s1 = fse_encode_sym(b, &w->wCt, s1, wt[i - 4]);
        i -= 4;
    }
    bw_add(b, s2, w->wCt.tableLog); bw_add(b, s1, w->wCt.tableLog);
    op = bw_close(b);
    return (uint32_t)(op - dst);
}

// huf_compress.c:248-289 HUF_writeCTable_wksp. dst: >= 132 bytes. returns size, 0 on failure
__device__ inline uint32_t huf_write_table(HufWork* w, uint8_t* dst, const uint32_t* code, uint32_t maxSym, uint32_t huffLog)
{
    for (uint32_t n = 0; n < maxSym; n++) { uint32_t const nb = code[n] & 0xFF; w->weights[n] = nb ? (uint8_t)(huffLog + 1 - nb) : 0; }
    {   uint32_t const h = huf_compress_weights(w, dst + 1, w->weights, maxSym);
        if (h > 1 && h < maxSym / 2) { dst[0] = (uint8_t)h; return h + 1; }
    }
    if (maxSym > 128) return 0;
    dst[0] = (uint8_t)(128 + (maxSym - 1));
    w->weights[maxSym] = 0;
    for (uint32_t n = 0; n < maxSym; n += 2) dst[n / 2 + 1] = (uint8_t)((w->weights[n] << 4) + w->weights[n + 1]);
    return ((maxSym + 1) / 2) + 1;
}

}  // namespace zhip
