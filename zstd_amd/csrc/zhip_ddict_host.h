// zhip_ddict_host.h — host side of the decoder: frame walking (what ZSTD_findFrameCompressedSize / ZSTD_getFrameContentSize
// compute, lib/decompress/zstd_decompress.c:590-850), the predefined FSE decoding tables, and a dictionary in decoding form
// (ZSTD_loadDEntropy + ZSTD_decompress_insertDictionary, zstd_decompress.c:1400-1500).  Plain C++, runs once per call /
// per dictionary; the per-frame work is all on the device (zhip_decode.h).
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>
#include "zhip_decode.h"
#include "zhip_cdict_host.h"

namespace zhip {

// LL[64] OF[32] ML[64]: the decoding tables of the predefined distributions (zstd_decompress_block.c:347-470)
static inline void host_dec_default_tables(uint64_t out[160])
{
    for (int k = 0; k < 3; k++) {
        int16_t norm[64]; uint8_t symOf[64]; uint16_t next[64];
        uint32_t const maxSym = k == 0 ? 35u : k == 1 ? 28u : 52u, log = k == 1 ? 5u : 6u;
        for (uint32_t s = 0; s <= maxSym; s++) norm[s] = (int16_t)dec_default_norm(k, s);
        fse_d_build(out + (k == 0 ? 0 : k == 1 ? 64 : 96), symOf, next, norm, maxSym, k, log);
    }
}

// One frame's extent: returns 0 ok, else a zstd error code.  *cSize = its compressed size, *content = the content size
// stated by the header or ~0ull, *bound = nbBlocks * blockSizeMax (ZSTD_decompressBound's per-frame term)
static inline int host_frame_extent(const uint8_t* src, size_t n, size_t* cSize, uint64_t* content, uint64_t* bound)
{
    static const unsigned did[4] = { 0, 1, 2, 4 }, fcsB[4] = { 0, 2, 4, 8 };
    if (n < 5) return 72;
    uint32_t magic; memcpy(&magic, src, 4);
    if (magic != 0xFD2FB528u) return 10;
    unsigned const fhd = src[4], single = (fhd >> 5) & 1, fcsCode = fhd >> 6;
    if (fhd & 8) return 14;
    size_t pos = 5; uint64_t window = 0;
    size_t const hs = 5 + !single + did[fhd & 3] + fcsB[fcsCode] + (single && !fcsCode);
    if (hs + 3 > n) return 72;
    if (!single) { unsigned const wl = (src[pos] >> 3) + 10; if (wl > 31) return 16; window = 1ull << wl; window += (window >> 3) * (src[pos] & 7); pos++; }
    pos += did[fhd & 3];
    uint64_t fcs = ~0ull;
    {   unsigned const nb = fcsB[fcsCode] + (single && !fcsCode);
        if (nb) { fcs = 0; for (unsigned i = 0; i < nb; i++) fcs |= (uint64_t)src[pos + i] << (8 * i); if (fcsCode == 1) fcs += 256; }
        pos += nb;
    }
    if (single) window = fcs;
    uint64_t const blockMax = window < 131072 ? window : 131072;
    uint64_t blocks = 0;
    for (;;) {
        if (pos + 3 > n) return 72;
        uint32_t const bh = src[pos] | (src[pos + 1] << 8) | ((uint32_t)src[pos + 2] << 16);
        pos += 3;
        unsigned const type = (bh >> 1) & 3;
        if (type == 3) return 20;
        size_t const cs = type == 1 ? 1 : (bh >> 3);
        if (pos + cs > n) return 72;
        pos += cs; blocks++;
        if (bh & 1) break;
    }
    if (fhd & 4) { if (pos + 4 > n) return 72; pos += 4; }
    *cSize = pos; *content = fcs; *bound = blocks * blockMax;
    return 0;
}

// a dictionary in decoding form
struct HostDDict {
    std::vector<uint8_t> content;            // raw content (what virtually precedes every frame)
    uint32_t dictID = 0, hasEntropy = 0, hufLog = 0;
    std::vector<uint16_t> huf;               // 4096 entries
    std::vector<uint32_t> huf2;              // 2048 entries: the double-symbol form when hufLog <= 11
    std::vector<uint64_t> fse;               // LL[512] OF[256] ML[512]
    uint32_t log[3] = {0, 0, 0}, rep[3] = {1, 4, 8};
};

// tree description -> single-symbol decoding table (HUF_readStats + HUF_readDTableX1_wksp). returns bytes consumed, 0 on error
static inline size_t host_huf_dtable(std::vector<uint16_t>& T, uint32_t* logOut, const uint8_t* src, size_t size)
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
    T.assign(4096, 0);
    uint32_t start[14], pos = 0;
    for (unsigned r = 1; r <= tableLog; r++) { start[r] = pos; pos += rank[r] << (r - 1); }
    for (unsigned n = 0; n <= oSize; n++) if (w[n]) {
        uint32_t const len = (1u << w[n]) >> 1;
        for (uint32_t k = 0; k < len; k++) T[start[w[n]] + k] = (uint16_t)(n | ((tableLog + 1 - w[n]) << 8));
        start[w[n]] += len;
    }
    *logOut = tableLog;
    return iSize + 1;
}

// returns 0 ok, else a zstd error code (30 dictionary_corrupted)
static inline int host_ddict_build(HostDDict& d, const void* dictv, size_t dictSize)
{
    const uint8_t* const dict = (const uint8_t*)dictv;
    d.fse.assign(1280, 0); d.huf.assign(4096, 0); d.huf2.assign(2048, 0);
    uint32_t magic = 0; if (dictSize >= 8) memcpy(&magic, dict, 4);
    if (dictSize < 8 || magic != 0xEC30A437u) { d.content.assign(dict, dict + dictSize); return 0; }       // raw-content dictionary
    memcpy(&d.dictID, dict + 4, 4);
    const uint8_t* p = dict + 8; const uint8_t* const end = dict + dictSize;
    size_t h = host_huf_dtable(d.huf, &d.hufLog, p, (size_t)(end - p)); if (!h) return 30; p += h;
    for (int step = 0; step < 3; step++) {                       // order in the dictionary: OF, ML, LL (zstd_decompress.c:1425-1455)
        int const k = step == 0 ? 1 : step == 1 ? 2 : 0;
        int16_t norm[64]; uint8_t symOf[512]; uint16_t next[64]; unsigned maxSym = dec_max_sym(k), tl = 0;
        h = host_read_ncount(norm, &maxSym, &tl, p, (size_t)(end - p));
        if (!h || tl > dec_max_log(k)) return 30;
        fse_d_build(d.fse.data() + (k == 0 ? 0 : k == 1 ? 512 : 768), symOf, next, norm, maxSym, k, tl);
        d.log[k] = tl; p += h;
    }
    if (p + 12 > end) return 30;
    memcpy(d.rep, p, 12); p += 12;
    size_t const contentSize = (size_t)(end - p);
    for (int i = 0; i < 3; i++) if (d.rep[i] == 0 || d.rep[i] > contentSize) return 30;
    d.huf2.assign(2048, 0);
    if (d.hufLog <= 11) for (uint32_t i = 0; i < (1u << d.hufLog); i++) d.huf2[i] = huf_double_entry(d.huf.data(), d.hufLog, i);
    d.content.assign(p, end);
    d.hasEntropy = 1;
    return 0;
}

}  // namespace zhip
