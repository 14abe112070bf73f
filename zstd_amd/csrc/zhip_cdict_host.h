// zhip_cdict_host.h — host side of the dictionary path: what ZSTD_createCDict does once per dictionary
// (lib/compress/zstd_compress.c:5477-5660): pick the CDict's own parameters, index the dictionary content into tagged
// ("short cache") hash tables, and decide per record whether / with which working parameters the reference would attach it.
// Plain C++ (no HIP): the tables are uploaded by zhip_lib.hip and read by zhip_parse_dict.h.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "zhip_host.h"

namespace zhip {

struct HostCDict {
    std::vector<uint8_t> content;       // dictionary content + 16 zero bytes of padding
    size_t len;                         // content bytes (0: dictionaries below 8 bytes are ignored, zstd_compress.c:5130)
    CParams cp;                         // the CDict's parameters (ZSTD_cpm_createCDict)
    std::vector<uint32_t> tabL, tabS;   // index << 8 | tag; fast: tabL only
    uint32_t dictID; uint32_t rep[3];
    int level;
};

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

// returns 0 ok, 1 unsupported parameters (strategy above dfast), 2 unsupported dictionary format (ZDICT entropy tables)
static inline int host_cdict_build(HostCDict& cd, const void* dict, size_t dictSize, int level)
{
    if (!host_get_cparams_mode(level, HOST_SRCSIZE_UNKNOWN, dictSize, HOST_CPM_CREATE_CDICT, &cd.cp) || cd.cp.strategy > 2) return 1;
    cd.level = level == 0 ? 3 : level;
    if (dictSize >= 4) { uint32_t magic; memcpy(&magic, dict, 4); if (magic == 0xEC30A437U) return 2; }
    if (dictSize < 8) dictSize = 0;
    cd.len = dictSize;
    cd.content.assign(dictSize + 32, 0);
    if (dictSize) memcpy(cd.content.data(), dict, dictSize);
    cd.tabL.assign((size_t)1 << cd.cp.hashLog, 0);
    cd.tabS.assign(cd.cp.strategy == 2 ? (size_t)1 << cd.cp.chainLog : 1, 0);
    cd.dictID = 0; cd.rep[0] = 1; cd.rep[1] = 4; cd.rep[2] = 8;
    host_cdict_fill(cd);
    return 0;
}

// working-context parameters for a record of n bytes with `cd` attached (zstd_compress.c:6289-6292, :2318-2338);
// false when the reference would copy the dictionary instead of attaching it (:2289-2315) — not implemented
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
