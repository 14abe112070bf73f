/* oracle/zoracle_dec.c — TEST INFRASTRUCTURE ONLY (never linked by the product).
 *
 * Plain-C, single-threaded restatement of the reference's frame DECODER (facebook/zstd @ /root/reference,
 * lib/decompress + lib/common), the checker for the device decoder (zstd_amd/csrc/zhip_decode.h).  Pinned against the
 * real reference (ZSTD_decompress) by tests/test_oracle_decode.py: frames of every level, multi-block frames, the
 * reference's own golden-decompression fixtures and its golden-decompression-errors.  Corrupted input: never accepts what
 * the reference rejects; it is stricter than the reference's Huffman fast loop, which does not check exact consumption.
 *
 * What is restated (file:line of the reference):
 *   frame header            lib/decompress/zstd_decompress.c:438-545  ZSTD_getFrameHeader_advanced
 *   frame / block loop      lib/decompress/zstd_decompress.c:951-1064 ZSTD_decompressFrame, :1068 ZSTD_decompressMultiFrame
 *   block header            lib/decompress/zstd_decompress_block.c:71-90 ZSTD_getcBlockSize
 *   literals section        zstd_decompress_block.c:134-345 ZSTD_decodeLiteralsBlock
 *   Huffman tree + decode   lib/common/entropy_common.c:236-320 HUF_readStats, lib/decompress/huf_decompress.c:385-500
 *                           HUF_readDTableX1_wksp, :560-640 HUF_decompress1X1 / 4X1 (X2 decodes the same symbols)
 *   FSE table description   lib/common/entropy_common.c:42-214 FSE_readNCount, lib/common/fse_decompress.c:58-277
 *   sequence tables         zstd_decompress_block.c:484-585 ZSTD_buildFSETable, :625-660 ZSTD_buildSeqTable,
 *                           :662-745 ZSTD_decodeSeqHeaders; base/bits tables :347-470 and lib/common/zstd_internal.h:123-160
 *   sequence decoding       zstd_decompress_block.c:1228-1345 ZSTD_decodeSequence, :1615-1690 ZSTD_decompressSequences_body
 *   sequence execution      zstd_decompress_block.c:1001-1095 ZSTD_execSequence
 *   dictionary              lib/decompress/zstd_decompress.c:1400-1500 ZSTD_loadDEntropy / ZSTD_decompress_insertDictionary
 *   checksum                XXH64 low 32 bits (zstd_decompress.c:1047-1057)
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#define ZD_ERR ((size_t)-1)
#define ZD_BLOCK_MAX 131072u

uint64_t zo_xxh64(const void* src, size_t n, uint64_t seed);   /* zoracle.c */

static unsigned zd_hb(uint32_t v) { return 31u - (unsigned)__builtin_clz(v); }
static uint32_t zd_le(const uint8_t* p, unsigned n) { uint32_t v = 0; unsigned i; for (i = 0; i < n; i++) v |= (uint32_t)p[i] << (8 * i); return v; }

/* ------------------------------------------------------------------ backward bit reader (lib/common/bitstream.h:250-420)
 * The stream's last byte holds the end mark (its highest set bit); data bits lie below it and are consumed from the top
 * down.  `pos` = number of data bits not yet consumed; reading below bit 0 yields zeros and drives pos negative, which is
 * what BIT_endOfDStream's "exactly consumed" test turns into corruption_detected. */
typedef struct { const uint8_t* p; long pos; int bad; } zd_bits;
static int zd_bits_init(zd_bits* b, const uint8_t* p, size_t n)
{
    b->p = p; b->bad = 0; b->pos = 0;
    if (n == 0 || p[n - 1] == 0) { b->bad = 1; return -1; }           /* bitstream.h:262, :284 */
    b->pos = (long)(8 * (n - 1) + zd_hb(p[n - 1]));
    return 0;
}
static uint32_t zd_bits_peek(const zd_bits* b, unsigned n)            /* the n bits just below pos, as a number */
{
    uint32_t v = 0; unsigned k;
    for (k = 0; k < n; k++) { long const q = b->pos - 1 - (long)k; v <<= 1; if (q >= 0) v |= (b->p[q >> 3] >> (q & 7)) & 1u; }
    return v;
}
static uint32_t zd_bits_read(zd_bits* b, unsigned n) { uint32_t const v = zd_bits_peek(b, n); b->pos -= (long)n; return v; }

/* ------------------------------------------------------------------ FSE_readNCount (entropy_common.c:42-214) */
static uint32_t zd_fpeek(const uint8_t* p, size_t size, size_t bit, unsigned n)
{
    uint64_t v = 0; size_t const byte = bit >> 3; unsigned i;
    for (i = 0; i < 8; i++) if (byte + i < size) v |= (uint64_t)p[byte + i] << (8 * i);
    return (uint32_t)((v >> (bit & 7)) & ((1ULL << n) - 1));
}
static size_t zd_read_ncount(short* norm, unsigned* maxSym, unsigned* tableLog, const uint8_t* src, size_t size)
{
    size_t bit = 0; int remaining, threshold, nbBits; unsigned charnum = 0, maxSV1 = *maxSym + 1; int previous0 = 0;
    if (size == 0) return ZD_ERR;
    memset(norm, 0, sizeof(short) * maxSV1);
    nbBits = (int)zd_fpeek(src, size, bit, 4) + 5; bit += 4;
    if (nbBits > 15) return ZD_ERR;                                  /* FSE_TABLELOG_ABSOLUTE_MAX */
    *tableLog = (unsigned)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    while (remaining > 1 && charnum < maxSV1) {
        if (previous0) {
            for (;;) { uint32_t const r = zd_fpeek(src, size, bit, 2); bit += 2; charnum += r; if (r != 3) break; if (bit > 8 * size + 64) return ZD_ERR; }
            if (charnum >= maxSV1) break;
        }
        {   int const max = (2 * threshold - 1) - remaining;
            uint32_t const bits = zd_fpeek(src, size, bit, (unsigned)nbBits);
            int count;
            if ((int)(bits & (uint32_t)(threshold - 1)) < max) { count = (int)(bits & (uint32_t)(threshold - 1)); bit += (size_t)nbBits - 1; }
            else { count = (int)(bits & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= max; bit += (size_t)nbBits; }
            count--;
            remaining -= count < 0 ? -count : count;
            norm[charnum++] = (short)count;
            previous0 = !count;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
    }
    if (remaining != 1 || charnum > maxSV1) return ZD_ERR;
    if (bit > 8 * size) return ZD_ERR;                               /* :200 srcSize_wrong */
    *maxSym = charnum - 1;
    return (bit + 7) >> 3;
}

/* ------------------------------------------------------------------ Huffman */
typedef struct { uint8_t sym[4096]; uint8_t nb[4096]; unsigned tableLog; int valid; } zd_huf;

/* FSE-compressed weights: fse_decompress.c:58-277 (two interleaved states, table log <= 6) */
static size_t zd_fse_weights(uint8_t* dst, size_t cap, const uint8_t* src, size_t size)
{
    short norm[256]; unsigned maxSym = 255, tl; size_t const h = zd_read_ncount(norm, &maxSym, &tl, src, size);
    uint8_t symT[64], nbT[64]; uint16_t newT[64]; unsigned next[256];
    zd_bits b; size_t n = 0; unsigned s1, s2; int which = 0;
    if (h == ZD_ERR || tl > 6 || h >= size) return ZD_ERR;
    {   unsigned const tsz = 1u << tl, mask = tsz - 1, step = (tsz >> 1) + (tsz >> 3) + 3; unsigned high = tsz - 1, pos = 0, sy, u;
        for (sy = 0; sy <= maxSym; sy++) { if (norm[sy] == -1) { symT[high--] = (uint8_t)sy; next[sy] = 1; } else next[sy] = (unsigned)norm[sy]; }
        for (sy = 0; sy <= maxSym; sy++) { int i; for (i = 0; i < norm[sy]; i++) { symT[pos] = (uint8_t)sy; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; } }
        if (pos != 0) return ZD_ERR;
        for (u = 0; u < tsz; u++) { unsigned const ns = next[symT[u]]++; nbT[u] = (uint8_t)(tl - zd_hb(ns)); newT[u] = (uint16_t)((ns << nbT[u]) - tsz); }
    }
    if (zd_bits_init(&b, src + h, size - h)) return ZD_ERR;
    s1 = zd_bits_read(&b, tl); s2 = zd_bits_read(&b, tl);
    for (;;) {                                                        /* fse_decompress.c:207-233 tail loop */
        unsigned* const st = which ? &s2 : &s1; unsigned const other = which ? s1 : s2;
        unsigned const nb = nbT[*st];
        if (n + 2 > cap) return ZD_ERR;
        dst[n++] = symT[*st];
        *st = newT[*st] + zd_bits_read(&b, nb);
        if (b.pos < 0) { dst[n++] = symT[other]; break; }
        which ^= 1;
    }
    return n;
}

/* HUF_readStats + HUF_readDTableX1_wksp: tree description -> single-symbol decoding table. returns bytes consumed */
static size_t zd_huf_read(zd_huf* h, const uint8_t* src, size_t size)
{
    uint8_t w[256]; unsigned rank[16], nbSym, tableLog, n; size_t iSize, oSize; uint32_t total = 0;
    h->valid = 0;
    if (!size) return ZD_ERR;
    iSize = src[0];
    if (iSize >= 128) {
        oSize = iSize - 127; iSize = (oSize + 1) / 2;
        if (iSize + 1 > size || oSize >= 256) return ZD_ERR;
        for (n = 0; n < oSize; n += 2) { w[n] = src[1 + n / 2] >> 4; w[n + 1] = src[1 + n / 2] & 15; }
    } else {
        if (iSize + 1 > size) return ZD_ERR;
        oSize = zd_fse_weights(w, 255, src + 1, iSize);
        if (oSize == ZD_ERR) return ZD_ERR;
    }
    memset(rank, 0, sizeof(rank));
    for (n = 0; n < oSize; n++) { if (w[n] > 12) return ZD_ERR; rank[w[n]]++; total += (1u << w[n]) >> 1; }
    if (!total) return ZD_ERR;
    tableLog = zd_hb(total) + 1;
    if (tableLog > 12) return ZD_ERR;
    {   uint32_t const rest = (1u << tableLog) - total; unsigned const last = zd_hb(rest) + 1;
        if ((1u << zd_hb(rest)) != rest) return ZD_ERR;
        w[oSize] = (uint8_t)last; rank[last]++;
    }
    if (rank[1] < 2 || (rank[1] & 1)) return ZD_ERR;
    nbSym = (unsigned)oSize + 1;
    {   /* huf_decompress.c:427-480: symbols in weight order then symbol order fill consecutive table ranges, weight 1 first */
        unsigned start[14], wv; uint32_t pos = 0;
        for (wv = 1; wv <= tableLog; wv++) { start[wv] = pos; pos += rank[wv] << (wv - 1); }
        for (n = 0; n < nbSym; n++) if (w[n]) {
            uint32_t const len = (1u << w[n]) >> 1; uint32_t k;
            for (k = 0; k < len; k++) { h->sym[start[w[n]] + k] = (uint8_t)n; h->nb[start[w[n]] + k] = (uint8_t)(tableLog + 1 - w[n]); }
            start[w[n]] += len;
        }
    }
    h->tableLog = tableLog; h->valid = 1;
    return iSize + 1;
}

/* HUF_decompress1X1_usingDTable_internal_body (huf_decompress.c:560-590): n symbols, stream must end exactly */
static int zd_huf_1x(uint8_t* dst, size_t n, const uint8_t* src, size_t size, const zd_huf* h)
{
    zd_bits b; size_t i;
    if (zd_bits_init(&b, src, size)) return -1;
    for (i = 0; i < n; i++) { uint32_t const idx = zd_bits_peek(&b, h->tableLog); dst[i] = h->sym[idx]; b.pos -= h->nb[idx]; }
    return b.pos == 0 ? 0 : -1;
}
/* HUF_decompress4X1_usingDTable_internal_body (:600-700) */
static int zd_huf_4x(uint8_t* dst, size_t n, const uint8_t* src, size_t size, const zd_huf* h)
{
    size_t l1, l2, l3, l4, seg;
    if (size < 10 || n < 6) return -1;                                 /* :612, :613 */
    l1 = zd_le(src, 2); l2 = zd_le(src + 2, 2); l3 = zd_le(src + 4, 2);
    if (6 + l1 + l2 + l3 > size) return -1;
    l4 = size - 6 - l1 - l2 - l3;
    seg = (n + 3) / 4;
    if (3 * seg > n) return -1;                                        /* :634 opStart4 > oend */
    if (zd_huf_1x(dst, seg, src + 6, l1, h)) return -1;
    if (zd_huf_1x(dst + seg, seg, src + 6 + l1, l2, h)) return -1;
    if (zd_huf_1x(dst + 2 * seg, seg, src + 6 + l1 + l2, l3, h)) return -1;
    if (zd_huf_1x(dst + 3 * seg, n - 3 * seg, src + 6 + l1 + l2 + l3, l4, h)) return -1;
    return 0;
}

/* ------------------------------------------------------------------ sequence tables */
typedef struct { uint16_t next; uint8_t nbAdd; uint8_t nb; uint32_t base; } zd_sym;        /* = ZSTD_seqSymbol */
typedef struct { zd_sym t[512]; unsigned log; int valid; } zd_tab;

static const uint32_t ZD_LL_base[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
static const uint8_t  ZD_LL_bits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const uint32_t ZD_ML_base[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };
static const uint8_t  ZD_ML_bits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
static const short ZD_LL_def[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };     /* zstd_internal.h:150-160, log 6 */
static const short ZD_ML_def[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };   /* log 6 */
static const short ZD_OF_def[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };                 /* log 5 */

static uint32_t zd_of_base(unsigned c) { return c == 0 ? 0 : c == 1 ? 1 : (1u << c) - 3; }                             /* OF_base, zstd_decompress_block.c:395 */

/* kind: 0 LL, 1 OF, 2 ML */
static void zd_base_bits(int kind, unsigned s, uint32_t* base, uint8_t* bits)
{
    if (kind == 0) { *base = ZD_LL_base[s]; *bits = ZD_LL_bits[s]; }
    else if (kind == 1) { *base = zd_of_base(s); *bits = (uint8_t)s; }
    else { *base = ZD_ML_base[s]; *bits = ZD_ML_bits[s]; }
}
/* ZSTD_buildFSETable_body (zstd_decompress_block.c:484-585) */
static void zd_build(zd_tab* T, const short* norm, unsigned maxSym, int kind, unsigned tableLog)
{
    unsigned const tsz = 1u << tableLog, mask = tsz - 1, step = (tsz >> 1) + (tsz >> 3) + 3;
    unsigned high = tsz - 1, pos = 0, s, u; uint16_t next[64]; uint8_t symOf[512];
    for (s = 0; s <= maxSym; s++) { if (norm[s] == -1) { symOf[high--] = (uint8_t)s; next[s] = 1; } else next[s] = (uint16_t)norm[s]; }
    for (s = 0; s <= maxSym; s++) { int i; for (i = 0; i < norm[s]; i++) { symOf[pos] = (uint8_t)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; } }
    for (u = 0; u < tsz; u++) {
        unsigned const sy = symOf[u]; unsigned const ns = next[sy]++;
        T->t[u].nb = (uint8_t)(tableLog - zd_hb(ns));
        T->t[u].next = (uint16_t)((ns << T->t[u].nb) - tsz);
        zd_base_bits(kind, sy, &T->t[u].base, &T->t[u].nbAdd);
    }
    T->log = tableLog; T->valid = 1;
}
/* ZSTD_buildSeqTable (:625-660). returns header bytes consumed */
static size_t zd_seq_table(zd_tab* T, int type, int kind, const uint8_t* src, size_t size, int repeatOk)
{
    static const unsigned maxOf[3] = { 35, 31, 52 }, maxLog[3] = { 9, 8, 9 }, defLog[3] = { 6, 5, 6 };
    if (type == 0) {                                                   /* predefined */
        zd_build(T, kind == 0 ? ZD_LL_def : kind == 1 ? ZD_OF_def : ZD_ML_def, kind == 0 ? 35 : kind == 1 ? 28 : 52, kind, defLog[kind]);
        return 0;
    }
    if (type == 1) {                                                   /* RLE: one symbol, 0 state bits */
        if (!size || src[0] > maxOf[kind]) return ZD_ERR;
        T->t[0].nb = 0; T->t[0].next = 0; zd_base_bits(kind, src[0], &T->t[0].base, &T->t[0].nbAdd);
        T->log = 0; T->valid = 1;
        return 1;
    }
    if (type == 3) return (repeatOk && T->valid) ? 0 : ZD_ERR;
    {   short norm[64]; unsigned max = maxOf[kind], tl; size_t const h = zd_read_ncount(norm, &max, &tl, src, size);
        if (h == ZD_ERR || tl > maxLog[kind]) return ZD_ERR;
        zd_build(T, norm, max, kind, tl);
        return h;
    }
}

/* ------------------------------------------------------------------ decoder state for one frame */
typedef struct {
    zd_huf huf; zd_tab ll, of, ml; int fseValid;       /* dctx->litEntropy = huf.valid, dctx->fseEntropy = fseValid */
    uint32_t rep[3];
    const uint8_t* dict; size_t dictLen;               /* content that virtually precedes the frame (may be NULL) */
    uint8_t lit[ZD_BLOCK_MAX + 32];
} zd_state;

/* one compressed block (ZSTD_decompressBlock_internal, zstd_decompress_block.c:2072-2180). returns decoded size */
static size_t zd_block(zd_state* S, uint8_t* ostart, uint8_t* op, uint8_t* oend, const uint8_t* src, size_t size, size_t blockSizeMax)
{
    const uint8_t* ip = src; const uint8_t* const iend = src + size;
    const uint8_t* litPtr; size_t litSize;
    uint8_t* const obeg = op;
    if (size > blockSizeMax) return ZD_ERR;
    if (size < 2) return ZD_ERR;                                       /* MIN_CBLOCK_SIZE */
    {   unsigned const type = ip[0] & 3, sf = (ip[0] >> 2) & 3;
        if (type >= 2) {                                               /* compressed / treeless */
            size_t lh, cs; int single = 0; uint32_t lhc;
            if (type == 3 && !S->huf.valid) return ZD_ERR;
            if (size < 5) return ZD_ERR;
            lhc = zd_le(ip, 4);
            if (sf < 2) { single = !sf; lh = 3; litSize = (lhc >> 4) & 0x3FF; cs = (lhc >> 14) & 0x3FF; }
            else if (sf == 2) { lh = 4; litSize = (lhc >> 4) & 0x3FFF; cs = lhc >> 18; }
            else { lh = 5; litSize = (lhc >> 4) & 0x3FFFF; cs = (lhc >> 22) + ((size_t)ip[4] << 10); }
            if (litSize > blockSizeMax) return ZD_ERR;
            if (!single && litSize < 6) return ZD_ERR;                 /* MIN_LITERALS_FOR_4_STREAMS */
            if (cs + lh > size) return ZD_ERR;
            if (litSize > (size_t)(oend - op)) return ZD_ERR;               /* expectedWriteSize < litSize */
            {   const uint8_t* hs = ip + lh; size_t hn = cs;
                if (type == 2) {
                    size_t const t = zd_huf_read(&S->huf, hs, hn);
                    if (t == ZD_ERR || t >= hn) return ZD_ERR;         /* huf_decompress.c:938 hSize >= cSrcSize */
                    hs += t; hn -= t;
                }
                if (single ? zd_huf_1x(S->lit, litSize, hs, hn, &S->huf) : zd_huf_4x(S->lit, litSize, hs, hn, &S->huf)) return ZD_ERR;
            }
            litPtr = S->lit; ip += lh + cs;
        } else {
            size_t lh;
            if (sf == 0 || sf == 2) { lh = 1; litSize = ip[0] >> 3; }
            else if (sf == 1) { lh = 2; litSize = zd_le(ip, 2) >> 4; }
            else { lh = 3; if (size < 3) return ZD_ERR; litSize = zd_le(ip, 3) >> 4; }
            if (litSize > blockSizeMax) return ZD_ERR;
            if (type == 0) {
                if (lh + litSize > size) return ZD_ERR;
                litPtr = ip + lh; ip += lh + litSize;
            } else {
                if (lh + 1 > size) return ZD_ERR;
                memset(S->lit, ip[lh], litSize); litPtr = S->lit; ip += lh + 1;
            }
        }
    }
    /* sequences section (ZSTD_decodeSeqHeaders :662-745) */
    {   int nbSeq; const uint8_t* const litEnd = litPtr + litSize;
        if (ip >= iend) return ZD_ERR;                                 /* MIN_SEQUENCES_SIZE */
        nbSeq = *ip++;
        if (nbSeq > 0x7F) {
            if (nbSeq == 0xFF) { if (ip + 2 > iend) return ZD_ERR; nbSeq = (int)zd_le(ip, 2) + 0x7F00; ip += 2; }
            else { if (ip >= iend) return ZD_ERR; nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
        }
        if (nbSeq == 0) { if (ip != iend) return ZD_ERR; }
        else {
            unsigned modes; size_t h; zd_bits b; uint32_t sLL, sOF, sML; int i;
            uint32_t rep[3];
            if (ip + 1 > iend) return ZD_ERR;
            modes = *ip++;
            if (modes & 3) return ZD_ERR;
            h = zd_seq_table(&S->ll, (int)(modes >> 6), 0, ip, (size_t)(iend - ip), S->fseValid); if (h == ZD_ERR) return ZD_ERR; ip += h;
            h = zd_seq_table(&S->of, (int)((modes >> 4) & 3), 1, ip, (size_t)(iend - ip), S->fseValid); if (h == ZD_ERR) return ZD_ERR; ip += h;
            h = zd_seq_table(&S->ml, (int)((modes >> 2) & 3), 2, ip, (size_t)(iend - ip), S->fseValid); if (h == ZD_ERR) return ZD_ERR; ip += h;
            S->fseValid = 1;
            rep[0] = S->rep[0]; rep[1] = S->rep[1]; rep[2] = S->rep[2];
            if (zd_bits_init(&b, ip, (size_t)(iend - ip))) return ZD_ERR;
            sLL = zd_bits_read(&b, S->ll.log); sOF = zd_bits_read(&b, S->of.log); sML = zd_bits_read(&b, S->ml.log);
            for (i = 0; i < nbSeq; i++) {                              /* ZSTD_decodeSequence :1228-1345 */
                const zd_sym* const eL = &S->ll.t[sLL]; const zd_sym* const eO = &S->of.t[sOF]; const zd_sym* const eM = &S->ml.t[sML];
                uint64_t offset; size_t ml = eM->base, ll = eL->base;
                if (eO->nbAdd > 1) {
                    offset = (uint64_t)eO->base + zd_bits_read(&b, eO->nbAdd);
                    rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = (uint32_t)offset;
                } else {
                    unsigned const ll0 = (eL->base == 0);
                    if (eO->nbAdd == 0) { offset = rep[ll0]; rep[1] = rep[!ll0]; rep[0] = (uint32_t)offset; }
                    else {
                        uint32_t const v = eO->base + ll0 + zd_bits_read(&b, 1);
                        uint32_t t = (v == 3) ? rep[0] - 1 : rep[v];
                        if (t == 0) t = 0xFFFFFFFFu;                   /* 0 is invalid: forces corruption at execution */
                        if (v != 1) rep[2] = rep[1];
                        rep[1] = rep[0]; rep[0] = t; offset = t;
                    }
                }
                ml += zd_bits_read(&b, eM->nbAdd);
                ll += zd_bits_read(&b, eL->nbAdd);
                if (i + 1 < nbSeq) {                                    /* states: LL, ML, OF */
                    sLL = eL->next + zd_bits_read(&b, eL->nb);
                    sML = eM->next + zd_bits_read(&b, eM->nb);
                    sOF = eO->next + zd_bits_read(&b, eO->nb);
                }
                /* ZSTD_execSequence :1001-1095 */
                if (ll > (size_t)(litEnd - litPtr)) return ZD_ERR;
                if (ll + ml > (size_t)(oend - op)) return ZD_ERR;
                memcpy(op, litPtr, ll); op += ll; litPtr += ll;
                {   size_t const have = (size_t)(op - ostart);
                    size_t k = 0;
                    if (offset > have) {
                        size_t const back = (size_t)(offset - have);     /* bytes before the frame start = in the dictionary */
                        if (back > S->dictLen) return ZD_ERR;
                        {   const uint8_t* m = S->dict + S->dictLen - back;
                            while (k < ml && k < back) { op[k] = m[k]; k++; }
                        }
                    }
                    for (; k < ml; k++) op[k] = op[(ptrdiff_t)k - (ptrdiff_t)offset];
                    op += ml;
                }
            }
            if (b.pos != 0) return ZD_ERR;                              /* BIT_endOfDStream */
            S->rep[0] = rep[0]; S->rep[1] = rep[1]; S->rep[2] = rep[2];
        }
        {   size_t const last = (size_t)(litEnd - litPtr);
            if (last > (size_t)(oend - op)) return ZD_ERR;
            memcpy(op, litPtr, last); op += last;
        }
    }
    return (size_t)(op - obeg);
}

/* ZSTD_loadDEntropy (zstd_decompress.c:1400-1470): the entropy section of a ZDICT-format dictionary. returns its size */
static size_t zd_load_dict_entropy(zd_state* S, const uint8_t* dict, size_t dictSize)
{
    const uint8_t* p = dict + 8; const uint8_t* const end = dict + dictSize; size_t h; int k;
    if (dictSize <= 8) return ZD_ERR;
    h = zd_huf_read(&S->huf, p, (size_t)(end - p)); if (h == ZD_ERR) return ZD_ERR; p += h;
    {   short norm[64]; unsigned max, tl;
        max = 31; h = zd_read_ncount(norm, &max, &tl, p, (size_t)(end - p)); if (h == ZD_ERR || tl > 8) return ZD_ERR; zd_build(&S->of, norm, max, 1, tl); p += h;
        max = 52; h = zd_read_ncount(norm, &max, &tl, p, (size_t)(end - p)); if (h == ZD_ERR || tl > 9) return ZD_ERR; zd_build(&S->ml, norm, max, 2, tl); p += h;
        max = 35; h = zd_read_ncount(norm, &max, &tl, p, (size_t)(end - p)); if (h == ZD_ERR || tl > 9) return ZD_ERR; zd_build(&S->ll, norm, max, 0, tl); p += h;
    }
    if (p + 12 > end) return ZD_ERR;
    {   size_t const content = (size_t)(end - (p + 12));
        for (k = 0; k < 3; k++) { uint32_t const r = zd_le(p, 4); p += 4; if (r == 0 || r > content) return ZD_ERR; S->rep[k] = r; }
    }
    S->fseValid = 1;
    return (size_t)(p - dict);
}

/* One frame (ZSTD_decompressFrame).  *consumed = bytes of src it occupied.  dict may be NULL; raw-content or ZDICT format
 * (magic 0xEC30A437, zstd_decompress.c:1476-1500).  returns the decoded size or ZD_ERR */
size_t zo_decompress_frame_dict(void* dstv, size_t cap, const void* srcv, size_t n, size_t* consumed, const void* dictv, size_t dictSize)
{
    const uint8_t* const src = (const uint8_t*)srcv; const uint8_t* ip = src; size_t rem = n;
    uint8_t* const ostart = (uint8_t*)dstv; uint8_t* op = ostart; uint8_t* const oend = ostart + cap;
    unsigned fhd, dictIDCode, fcsCode, single, checksum; uint64_t fcs = (uint64_t)-1, windowSize = 0; uint32_t dictID = 0; size_t hs, blockSizeMax;
    zd_state* S; size_t result = ZD_ERR;
    if (consumed) *consumed = 0;
    if (n < 5 + 3) return ZD_ERR;                                      /* ZSTD_FRAMEHEADERSIZE_MIN + block header */
    if (zd_le(src, 4) != 0xFD2FB528u) return ZD_ERR;
    fhd = src[4]; dictIDCode = fhd & 3; checksum = (fhd >> 2) & 1; single = (fhd >> 5) & 1; fcsCode = fhd >> 6;
    if (fhd & 8) return ZD_ERR;                                        /* reserved bit: frameParameter_unsupported */
    {   static const unsigned did[4] = { 0, 1, 2, 4 }, fcsB[4] = { 0, 2, 4, 8 };
        hs = 5 + !single + did[dictIDCode] + fcsB[fcsCode] + (single && !fcsCode);
        if (n < hs + 3) return ZD_ERR;
        {   size_t pos = 5;
            if (!single) { unsigned const wl = (src[pos] >> 3) + 10; if (wl > 31) return ZD_ERR; windowSize = 1ULL << wl; windowSize += (windowSize >> 3) * (src[pos] & 7); pos++; }
            dictID = zd_le(src + pos, did[dictIDCode]); pos += did[dictIDCode];
            switch (fcsCode) {
                case 0: if (single) fcs = src[pos]; break;
                case 1: fcs = zd_le(src + pos, 2) + 256; break;
                case 2: fcs = zd_le(src + pos, 4); break;
                default: fcs = (uint64_t)zd_le(src + pos, 4) | ((uint64_t)zd_le(src + pos + 4, 4) << 32); break;
            }
            if (single) windowSize = fcs;
        }
    }
    blockSizeMax = windowSize < ZD_BLOCK_MAX ? (size_t)windowSize : ZD_BLOCK_MAX;
    ip += hs; rem -= hs;
    S = (zd_state*)calloc(1, sizeof(zd_state));
    S->rep[0] = 1; S->rep[1] = 4; S->rep[2] = 8;
    if (dictv && dictSize >= 8 && zd_le((const uint8_t*)dictv, 4) == 0xEC30A437u) {
        size_t const e = zd_load_dict_entropy(S, (const uint8_t*)dictv, dictSize);
        if (e == ZD_ERR) goto done;                                    /* dictionary_corrupted */
        if (dictID && dictID != zd_le((const uint8_t*)dictv + 4, 4)) goto done;   /* dictionary_wrong */
        S->dict = (const uint8_t*)dictv + e; S->dictLen = dictSize - e;
    } else {
        if (dictID) goto done;                                         /* frame asks for a dictionary we do not hold */
        S->dict = (const uint8_t*)dictv; S->dictLen = dictv ? dictSize : 0;
    }
    for (;;) {
        uint32_t bh; unsigned last, type; size_t bsize, csize, dec;
        if (rem < 3) goto done;
        bh = zd_le(ip, 3); last = bh & 1; type = (bh >> 1) & 3; bsize = bh >> 3;
        if (type == 3) goto done;
        csize = type == 1 ? 1 : bsize;
        ip += 3; rem -= 3;
        if (csize > rem) goto done;
        if (type == 2) { dec = zd_block(S, ostart, op, oend, ip, csize, blockSizeMax); if (dec == ZD_ERR) goto done; }
        else if (type == 0) { if (bsize > (size_t)(oend - op)) goto done; memcpy(op, ip, bsize); dec = bsize; }
        else { if (bsize > (size_t)(oend - op)) goto done; memset(op, ip[0], bsize); dec = bsize; }
        op += dec; ip += csize; rem -= csize;
        if (last) break;
    }
    if (fcs != (uint64_t)-1 && (uint64_t)(op - ostart) != fcs) goto done;
    if (checksum) {
        if (rem < 4) goto done;
        if (zd_le(ip, 4) != (uint32_t)zo_xxh64(ostart, (size_t)(op - ostart), 0)) goto done;
        ip += 4; rem -= 4;
    }
    if (consumed) *consumed = (size_t)(ip - src);
    result = (size_t)(op - ostart);
done:
    free(S);
    return result;
}

/* ZSTD_decompress (zstd_decompress.c:1068-1200 ZSTD_decompressMultiFrame): every frame of src, skippable frames skipped */
size_t zo_decompress_dict(void* dst, size_t cap, const void* srcv, size_t n, const void* dict, size_t dictSize)
{
    const uint8_t* src = (const uint8_t*)srcv; uint8_t* op = (uint8_t*)dst; size_t total = 0; int more = 0;
    while (n >= 5) {                                                    /* ZSTD_startingInputLength */
        size_t used, d;
        if (n >= 8 && (zd_le(src, 4) & 0xFFFFFFF0u) == 0x184D2A50u) {  /* skippable frame :1100 */
            size_t const sk = (size_t)zd_le(src + 4, 4) + 8;
            if (sk > n) return ZD_ERR;
            src += sk; n -= sk; continue;
        }
        d = zo_decompress_frame_dict(op, cap, src, n, &used, dict, dictSize);
        if (d == ZD_ERR) return ZD_ERR;
        op += d; cap -= d; total += d; src += used; n -= used; more = 1;
    }
    (void)more;
    if (n) return ZD_ERR;                                               /* :1195 srcSize_wrong: trailing garbage */
    return total;
}
size_t zo_decompress(void* dst, size_t cap, const void* src, size_t n) { return zo_decompress_dict(dst, cap, src, n, NULL, 0); }

/* frame boundaries without decoding (ZSTD_findFrameSizeInfo zstd_decompress.c:770-850): compressed size of the first frame
 * and its content size (or (uint64)-1 when the header does not hold it).  returns 0 ok, -1 error */
int zo_frame_info(const void* srcv, size_t n, size_t* compressedSize, unsigned long long* contentSize)
{
    const uint8_t* const src = (const uint8_t*)srcv; size_t pos; unsigned fhd, single, fcsCode, dictIDCode;
    static const unsigned did[4] = { 0, 1, 2, 4 }, fcsB[4] = { 0, 2, 4, 8 };
    if (n < 8 || zd_le(src, 4) != 0xFD2FB528u) return -1;
    fhd = src[4]; dictIDCode = fhd & 3; single = (fhd >> 5) & 1; fcsCode = fhd >> 6;
    pos = 5 + !single + did[dictIDCode];
    if (pos + fcsB[fcsCode] + (single && !fcsCode) > n) return -1;
    if (contentSize) {
        switch (fcsCode) {
            case 0: *contentSize = single ? src[pos] : (unsigned long long)-1; break;
            case 1: *contentSize = zd_le(src + pos, 2) + 256; break;
            case 2: *contentSize = zd_le(src + pos, 4); break;
            default: *contentSize = (unsigned long long)zd_le(src + pos, 4) | ((unsigned long long)zd_le(src + pos + 4, 4) << 32); break;
        }
    }
    pos += fcsB[fcsCode] + (single && !fcsCode);
    for (;;) {
        uint32_t bh; size_t cs;
        if (pos + 3 > n) return -1;
        bh = zd_le(src + pos, 3); pos += 3;
        if (((bh >> 1) & 3) == 3) return -1;
        cs = ((bh >> 1) & 3) == 1 ? 1 : (bh >> 3);
        if (pos + cs > n) return -1;
        pos += cs;
        if (bh & 1) break;
    }
    if (fhd & 4) { if (pos + 4 > n) return -1; pos += 4; }
    if (compressedSize) *compressedSize = pos;
    return 0;
}
