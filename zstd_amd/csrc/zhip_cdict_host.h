// zhip_cdict_host.h — host side of the dictionary path: what ZSTD_createCDict does once per dictionary
// (lib/compress/zstd_compress.c:5477-5660): pick the CDict's own parameters, index the dictionary content into tagged
// ("short cache") hash tables, and decide per record whether / with which working parameters the reference would attach it.
// Plain C++ (no HIP): the tables are uploaded by zhip_lib.hip and read by zhip_parse_dict.h.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "zhip_host.h"
#include "zhip_tables.h"

namespace zhip {

struct HostCDict {
    std::vector<uint8_t> content;       // dictionary content + 16 zero bytes of padding
    size_t len;                         // content bytes (0: dictionaries below 8 bytes are ignored, zstd_compress.c:5130)
    size_t fullSize;                    // the dictionary buffer as given (cdict->dictContentSize): sizes the window in copy mode
    CParams cp;                         // the CDict's parameters (ZSTD_cpm_createCDict)
    std::vector<uint32_t> tabL, tabS;   // index << 8 | tag; fast: tabL only
    uint32_t dictID; uint32_t rep[3];
    int level;
    bool hasEntropy; ZhipDictEntropy ent;   // ZDICT-format dictionaries
};

// ---- readers for the dictionary's entropy section (doc/zstd_compression_format.md "Dictionary Format")
static inline uint32_t host_bits_at(const uint8_t* p, size_t size, size_t bit, unsigned n)      // LSB-first, bytes past `size` read as 0
{
    uint64_t v = 0; size_t const byte = bit >> 3;
    for (unsigned i = 0; i < 8; i++) if (byte + i < size) v |= (uint64_t)p[byte + i] << (8 * i);
    return (uint32_t)((v >> (bit & 7)) & ((1ULL << n) - 1));
}
// FSE table description (lib/common/entropy_common.c:42-214 FSE_readNCount); returns bytes consumed, 0 on error
static inline size_t host_read_ncount(int16_t* norm, unsigned* maxSym, unsigned* tableLog, const uint8_t* src, size_t size)
{
    size_t bit = 0; unsigned const maxSV1 = *maxSym + 1; unsigned charnum = 0; bool previous0 = false;
    for (unsigned i = 0; i < maxSV1; i++) norm[i] = 0;
    int nbBits = (int)host_bits_at(src, size, bit, 4) + 5; bit += 4;
    if (nbBits > 15) return 0;
    *tableLog = (unsigned)nbBits;
    int remaining = (1 << nbBits) + 1, threshold = 1 << nbBits; nbBits++;
    while (remaining > 1 && charnum < maxSV1) {
        if (previous0) {
            for (;;) { uint32_t const r = host_bits_at(src, size, bit, 2); bit += 2; charnum += r; if (r != 3) break; }
            if (charnum >= maxSV1) break;
        }
        int const max = (2 * threshold - 1) - remaining;
        uint32_t const bits = host_bits_at(src, size, bit, (unsigned)nbBits);
        int count;
        if ((int)(bits & (uint32_t)(threshold - 1)) < max) { count = (int)(bits & (uint32_t)(threshold - 1)); bit += (size_t)nbBits - 1; }
        else { count = (int)(bits & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= max; bit += (size_t)nbBits; }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[charnum++] = (int16_t)count;
        previous0 = count == 0;
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
    }
    if (remaining != 1 || charnum > maxSV1 || bit > 8 * size) return 0;    // bits past the input read as zero: a table they complete is an overrun
    *maxSym = charnum - 1;
    return (bit + 7) >> 3;
}
static inline unsigned host_rbits(const uint8_t* bs, long at, unsigned nb)        // backward bitstream: nb bits whose lowest sits at `at`
{
    unsigned v = 0;
    for (unsigned k = 0; k < nb; k++) { long const q = at + (long)k; if (q >= 0 && ((bs[q >> 3] >> (q & 7)) & 1)) v |= 1u << k; }
    return v;
}
// FSE-compressed Huffman weights (lib/common/fse_decompress.c); returns the number of weights, 0 on error
static inline size_t host_fse_decode_weights(uint8_t* dst, size_t cap, const uint8_t* src, size_t size)
{
    int16_t norm[256]; unsigned maxSym = 255, tl;
    size_t const h = host_read_ncount(norm, &maxSym, &tl, src, size);
    if (!h || tl > 6 || h >= size) return 0;
    uint8_t symT[64], nbT[64]; uint16_t newT[64]; unsigned next[256];
    unsigned const tsz = 1u << tl, mask = tsz - 1, step = (tsz >> 1) + (tsz >> 3) + 3; unsigned high = tsz - 1, pos = 0;
    for (unsigned sy = 0; sy <= maxSym; sy++) { if (norm[sy] == -1) { symT[high--] = (uint8_t)sy; next[sy] = 1; } else next[sy] = (unsigned)norm[sy]; }
    for (unsigned sy = 0; sy <= maxSym; sy++) for (int i = 0; i < norm[sy]; i++) { symT[pos] = (uint8_t)sy; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; }
    if (pos != 0) return 0;
    for (unsigned u = 0; u < tsz; u++) { unsigned const ns = next[symT[u]]++; nbT[u] = (uint8_t)(tl - hb32(ns)); newT[u] = (uint16_t)((ns << nbT[u]) - tsz); }
    const uint8_t* const bs = src + h; size_t const bsz = size - h;
    if (bs[bsz - 1] == 0) return 0;
    long cursor = (long)(8 * (bsz - 1) + hb32(bs[bsz - 1]));
    cursor -= (long)tl; unsigned s1 = host_rbits(bs, cursor, tl);
    cursor -= (long)tl; unsigned s2 = host_rbits(bs, cursor, tl);
    size_t n = 0; bool second = false;
    for (;;) {
        unsigned& st = second ? s2 : s1; unsigned const other = second ? s1 : s2;
        if (n + 2 > cap) return 0;
        dst[n++] = symT[st];
        unsigned const nb = nbT[st];
        cursor -= (long)nb;
        st = newT[st] + host_rbits(bs, cursor, nb);
        if (cursor < 0) { dst[n++] = symT[other]; break; }
        second = !second;
    }
    return n;
}
// Huffman table description -> code (HUF_readStats + HUF_readCTable, entropy_common.c:248-320, huf_compress.c:291-339)
static inline size_t host_read_huf(ZhipDictEntropy& e, const uint8_t* src, size_t size)
{
    uint8_t w[256]; unsigned rank[16] = {0}; size_t iSize, oSize; uint32_t total = 0;
    if (!size) return 0;
    iSize = src[0];
    if (iSize >= 128) {
        oSize = iSize - 127; iSize = (oSize + 1) / 2;
        if (iSize + 1 > size || oSize >= 256) return 0;
        for (size_t n = 0; n < oSize; n += 2) { w[n] = src[1 + n / 2] >> 4; w[n + 1] = src[1 + n / 2] & 15; }
    } else {
        if (iSize + 1 > size) return 0;
        oSize = host_fse_decode_weights(w, 255, src + 1, iSize);
        if (!oSize) return 0;
    }
    for (size_t n = 0; n < oSize; n++) { if (w[n] > 12) return 0; rank[w[n]]++; total += (1u << w[n]) >> 1; }
    if (!total) return 0;
    unsigned const tableLog = hb32(total) + 1;
    if (tableLog > 12) return 0;
    {   uint32_t const rest = (1u << tableLog) - total; unsigned const last = hb32(rest) + 1;
        if ((1u << hb32(rest)) != rest) return 0;
        w[oSize] = (uint8_t)last; rank[last]++;
    }
    if (rank[1] < 2 || (rank[1] & 1)) return 0;
    unsigned const nbSym = (unsigned)oSize + 1;
    e.hufMaxSym = nbSym - 1;
    e.hufRepeat = (rank[0] == 0 && nbSym == 256) ? 2 : 1;
    uint8_t nbBits[256] = {0}; uint16_t nbPerRank[16] = {0}, valPerRank[16] = {0}; uint16_t min = 0;
    for (unsigned n = 0; n < nbSym; n++) { nbBits[n] = w[n] ? (uint8_t)(tableLog + 1 - w[n]) : 0; nbPerRank[nbBits[n]]++; }
    for (unsigned n = tableLog; n > 0; n--) { valPerRank[n] = min; min = (uint16_t)(min + nbPerRank[n]); min >>= 1; }
    for (unsigned n = 0; n < 256; n++) e.hufCode[n] = 0;
    for (unsigned n = 0; n < nbSym; n++) if (nbBits[n]) e.hufCode[n] = ((uint32_t)valPerRank[nbBits[n]]++ << 8) | nbBits[n];
    return iSize + 1;
}
static inline uint32_t host_ncount_repeat(const int16_t* norm, unsigned dictMax, unsigned maxSym)     // ZSTD_dictNCountRepeat :4966
{
    if (dictMax < maxSym) return 1;
    for (unsigned s = 0; s <= maxSym; s++) if (norm[s] == 0) return 1;
    return 2;
}
// the entropy section + repcodes of a ZDICT-format dictionary (ZSTD_loadCEntropy); returns its size, 0 if malformed
static inline size_t host_load_entropy(HostCDict& cd, const uint8_t* dict, size_t dictSize)
{
    const uint8_t* p = dict + 8; const uint8_t* const dEnd = dict + dictSize;
    int16_t ofN[32], mlN[53], llN[36]; unsigned ofMax = 31, ofLog, mlMax = 52, mlLog, llMax = 35, llLog; size_t h;
    memcpy(&cd.dictID, dict + 4, 4);
    h = host_read_huf(cd.ent, p, (size_t)(dEnd - p)); if (!h) return 0; p += h;
    h = host_read_ncount(ofN, &ofMax, &ofLog, p, (size_t)(dEnd - p)); if (!h || ofLog > 8) return 0; p += h;
    fse_build_ctable_host(&cd.ent.ct[1], ofN, 31, ofLog);                 // all offset symbols (MaxOff)
    h = host_read_ncount(mlN, &mlMax, &mlLog, p, (size_t)(dEnd - p)); if (!h || mlLog > 9) return 0; p += h;
    fse_build_ctable_host(&cd.ent.ct[2], mlN, mlMax, mlLog);
    cd.ent.fseRepeat[2] = host_ncount_repeat(mlN, mlMax, 52);
    h = host_read_ncount(llN, &llMax, &llLog, p, (size_t)(dEnd - p)); if (!h || llLog > 9) return 0; p += h;
    fse_build_ctable_host(&cd.ent.ct[0], llN, llMax, llLog);
    cd.ent.fseRepeat[0] = host_ncount_repeat(llN, llMax, 35);
    if (p + 12 > dEnd) return 0;
    memcpy(cd.rep, p, 12); p += 12;
    size_t const contentSize = (size_t)(dEnd - p);
    unsigned const offcodeMax = hb32((uint32_t)contentSize + 131072);
    cd.ent.fseRepeat[1] = host_ncount_repeat(ofN, ofMax, offcodeMax < 31 ? offcodeMax : 31);
    for (int i = 0; i < 3; i++) if (cd.rep[i] == 0 || cd.rep[i] > contentSize) return 0;
    return (size_t)(p - dict);
}

// ZSTD_hashPtr (zstd_compress_internal.h:820-862): top hBits of the multiplicative hash of the first mls bytes
static inline uint32_t host_hash(const uint8_t* p, unsigned hBits, unsigned mls)
{
    uint64_t v8; uint32_t v4;
    memcpy(&v4, p, 4);
    if (mls <= 4) return (v4 * 2654435761U) >> (32 - hBits);
    memcpy(&v8, p, 8);
    switch (mls) {
    case 5:  return (uint32_t)(((v8 << 24) * 889523592379ULL) >> (64 - hBits));
    case 6:  return (uint32_t)(((v8 << 16) * 227718039650203ULL) >> (64 - hBits));
    case 7:  return (uint32_t)(((v8 << 8) * 58295818150454627ULL) >> (64 - hBits));
    default: return (uint32_t)((v8 * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}
static inline void host_put_tagged(std::vector<uint32_t>& t, uint32_t hashAndTag, uint32_t index)   // ZSTD_writeTaggedIndex, internal.h:1404
{
    t[hashAndTag >> 8] = (index << 8) | (hashAndTag & 0xFF);
}

// dictionary byte j has index j + 2 (ZSTD_WINDOW_START_INDEX); ZSTD_fillHashTableForCDict (zstd_fast.c:16-49) and
// ZSTD_fillDoubleHashTableForCDict (zstd_double_fast.c:18-54), both with ZSTD_dtlm_full
static inline void host_cdict_fill(HostCDict& cd)
{
    const uint8_t* const base = cd.content.data() - 2;
    size_t const endIdx = cd.len + 2;
    unsigned const mls = cd.cp.minMatch;
    size_t first = 2;
    {   // a dictionary larger than the tables can index only has its suffix indexed (zstd_compress.c:4888-4896)
        unsigned const m = cd.cp.hashLog > cd.cp.chainLog ? cd.cp.hashLog : cd.cp.chainLog;
        size_t const maxDictSize = (size_t)8 << (m < 28 ? m : 28);
        if (cd.len > maxDictSize) first = 2 + (cd.len - maxDictSize);
    }
    if (endIdx - first <= 8) return;
    if (cd.cp.strategy == 1) {
        unsigned const hb = cd.cp.hashLog + 8;
        for (size_t ip = first; ip + 3 < (endIdx - 8) + 2; ip += 3) {
            host_put_tagged(cd.tabL, host_hash(base + ip, hb, mls), (uint32_t)ip);
            for (unsigned q = 1; q < 3; q++) {
                uint32_t const ht = host_hash(base + ip + q, hb, mls);
                if (cd.tabL[ht >> 8] == 0) host_put_tagged(cd.tabL, ht, (uint32_t)(ip + q));
            }
        }
    } else {
        unsigned const hbL = cd.cp.hashLog + 8, hbS = cd.cp.chainLog + 8;
        for (size_t ip = first; ip + 2 <= endIdx - 8; ip += 3) {
            for (unsigned i = 0; i < 3; i++) {
                uint32_t const sm = host_hash(base + ip + i, hbS, mls), lg = host_hash(base + ip + i, hbL, 8);
                if (i == 0) host_put_tagged(cd.tabS, sm, (uint32_t)(ip + i));
                if (i == 0 || cd.tabL[lg >> 8] == 0) host_put_tagged(cd.tabL, lg, (uint32_t)(ip + i));
            }
        }
    }
}

// returns 0 ok, 1 unsupported parameters (strategy above dfast), 2 malformed ZDICT-format dictionary
static inline int host_cdict_build(HostCDict& cd, const void* dict, size_t dictSize, int level)
{
    if (!host_get_cparams_mode(level, HOST_SRCSIZE_UNKNOWN, dictSize, HOST_CPM_CREATE_CDICT, &cd.cp) || cd.cp.strategy > 2) return 1;
    cd.level = level == 0 ? 3 : level;
    cd.fullSize = dictSize;
    cd.dictID = 0; cd.rep[0] = 1; cd.rep[1] = 4; cd.rep[2] = 8; cd.hasEntropy = false;
    memset(&cd.ent, 0, sizeof(cd.ent));
    const uint8_t* content = (const uint8_t*)dict;
    if (dictSize < 8) dictSize = 0;
    if (dictSize >= 8) {
        uint32_t magic; memcpy(&magic, dict, 4);
        if (magic == 0xEC30A437U) {                       // ZDICT format (ZSTD_loadZstdDictionary, zstd_compress.c:5087-5118)
            size_t const e = host_load_entropy(cd, content, dictSize);
            if (!e) return 2;
            content += e; dictSize -= e; cd.hasEntropy = true;
        }
    }
    cd.len = dictSize;
    cd.content.assign(dictSize + 32, 0);
    if (dictSize) memcpy(cd.content.data(), content, dictSize);
    cd.tabL.assign((size_t)1 << cd.cp.hashLog, 0);
    cd.tabS.assign(cd.cp.strategy == 2 ? (size_t)1 << cd.cp.chainLog : 1, 0);
    host_cdict_fill(cd);
    return 0;
}

// COPY mode (zstd_compress.c:2395-2419): sources above the attach cut-off.  The CDict's table parameters as they are, windowLog
// from the parameters requested for (level, srcSize, dictSize) with the dictionary counted in (ZSTD_cpm_noAttachDict, :6289-6292)
static inline bool host_cdict_copy_params(const HostCDict& cd, size_t n, CParams* out)
{
    CParams p, w;
    if (!host_get_cparams_mode(cd.level, n, cd.fullSize, HOST_CPM_NONE, &p)) return false;
    w = cd.cp; w.windowLog = p.windowLog;
    *out = w;
    return true;
}
static inline bool host_cdict_is_copy_mode(const HostCDict& cd, size_t n)
{
    static const size_t cutoff[3] = { 8192, 8192, 16384 };
    return n > cutoff[cd.cp.strategy];
}

// working-context parameters for a record of n bytes with `cd` attached (zstd_compress.c:6289-6292, :2318-2338);
// false when the reference would copy the dictionary instead of attaching it (:2289-2315): host_cdict_copy_params then
static inline bool host_cdict_unit_params(const HostCDict& cd, size_t n, CParams* out)
{
    static const size_t cutoff[3] = { 8192, 8192, 16384 };
    CParams p, w;
    if (n > cutoff[cd.cp.strategy]) return false;
    if (!host_get_cparams_mode(cd.level, n, cd.len, HOST_CPM_ATTACH, &p)) return false;
    w = cd.cp;
    host_adjust_cparams(&w, n, cd.len, HOST_CPM_ATTACH);
    w.windowLog = p.windowLog;
    *out = w;
    return true;
}

}  // namespace zhip
