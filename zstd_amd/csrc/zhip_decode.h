// zhip_decode.h — gfx950 frame decoder: the step on the other side of the block-compression path.
//
// WHAT it computes: for every frame of a batch exactly what the reference's ZSTD_decompress (lib/decompress/
// zstd_decompress.c:951-1064 ZSTD_decompressFrame) regenerates — any RFC 8878 frame (raw / RLE / compressed blocks,
// several blocks per frame with repeat / treeless modes, optional checksum, optional dictionary), i.e. the frames this
// library emits and the frames the reference emits.  Corrupted input is reported per frame with zstd's error codes.
//
// HOW (CDNA4 design).  Entropy decoding is serial per bitstream (a Huffman stream or the interleaved FSE stream cannot be
// entered in the middle), so parallelism comes from the batch: one 128-thread workgroup (two wavefronts) per frame, eight
// workgroups resident per CU (20 KB of LDS each), workgroups fetch frames from a queue.  Inside a workgroup the two
// wavefronts are a producer / consumer pair:
//   * wave 0: the literals section — Huffman table from the tree description into LDS (HUF_readStats + HUF_readDTableX1,
//     lib/common/entropy_common.c:236-320, lib/decompress/huf_decompress.c:385-500), then the four streams decoded by
//     four lanes in lockstep (one LDS lookup per symbol; the compressed bytes are streamed 8 at a time one load ahead so
//     the serial chain never waits on HBM);
//   * wave 1: the sequences section — the three FSE decoding tables into LDS (ZSTD_buildFSETable,
//     zstd_decompress_block.c:484-585; three lanes build LL / OF / ML concurrently), then one lane walks the interleaved
//     bitstream (ZSTD_decodeSequence :1228-1345) and hands over records {output position, literal position, offset,
//     match length} in chunks of ZHIP_DEC_CHUNK through an L2-resident buffer;
//   * wave 0 then executes a chunk (ZSTD_execSequence :1001-1095) while wave 1 decodes the next one: literals of 64
//     sequences are copied lane-parallel (their positions are known from the records), matches in order, each with a
//     wave-wide copy (512 B per step); a match whose source was written since the last `s_waitcnt vmcnt(0)` waits, all
//     others stream.  Overlapping matches (offset < length) are a periodic pattern of bytes that already exist, so they
//     are copied lane-parallel too (source index modulo the offset).
// Bound: latency of the serial chains (LDS lookups, one L2 round trip per dependent match); HBM traffic is the compressed
// bytes in, the content out, plus 32 B per sequence of hand-over records that stay in L2.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_parse.h"

#define ZHIP_DEC_THREADS 128
#ifndef ZHIP_DEC_CHUNK
#define ZHIP_DEC_CHUNK   1024u                 /* sequences per hand-over buffer */
#endif
#define ZHIP_DEC_LIT_STRIDE (ZHIP_UNIT_MAX + 64)
#define ZHIP_DEC_RING_WORDS 512u                /* dwords of the sequence bitstream staged in LDS (two halves) */

// zstd error codes (lib/zstd_errors.h) used as per-frame status
enum { ZHIP_DE_OK = 0, ZHIP_DE_PREFIX = 10, ZHIP_DE_UNSUPPORTED = 14, ZHIP_DE_WINDOW = 16, ZHIP_DE_CORRUPT = 20, ZHIP_DE_CHECKSUM = 22,
       ZHIP_DE_DICT_CORRUPT = 30, ZHIP_DE_DICT_WRONG = 32, ZHIP_DE_DST_SMALL = 70, ZHIP_DE_SRC_WRONG = 72 };

struct ZhipDFrame {          // one per frame, filled by the host
    uint64_t srcOff;         // where the frame starts in the compressed buffer
    uint64_t dstOff;         // where its content goes in the destination buffer
    uint32_t srcLen;         // compressed size of the frame (header .. checksum)
    uint32_t dstCap;         // room at dstOff (the content size when the frame header states it)
};
struct ZhipDResult {         // one per frame, written by the device
    uint32_t status;         // 0 or a zstd error code
    uint32_t size;           // decoded bytes
    uint32_t hasChecksum;    // frame carries a content checksum ...
    uint32_t checksum;       // ... this one (verified by k_xxh64 + k_dec_verify)
};
struct ZhipDSeq { uint32_t outPos, litPos, off, ml; };      // positions are frame- resp. block-relative

// a dictionary as the decoder needs it (host-built, zhip_ddict_host.h): content + the entropy tables in decoding form
struct ZhipDDictDev {
    const uint8_t*  content; uint32_t len; uint32_t dictID;
    uint32_t hasEntropy, hufLog;
    const uint16_t* huf;                     // 1 << hufLog entries: symbol | nbBits << 8
    const uint32_t* huf2;                    // hufLog <= 11: the same table in double-symbol form (huf_double_entry)
    const uint64_t* fse;                     // LL[512] OF[256] ML[512] in ZhipFseD packing
    uint32_t log[3];                         // LL, OF, ML table logs
    uint32_t rep[3];
};

namespace zhip {

// ------------------------------------------------------------------ tables of the format (doc/zstd_compression_format.md;
// lib/common/zstd_internal.h:123-160, lib/decompress/zstd_decompress_block.c:347-470)
__device__ __host__ inline uint32_t dec_ll_base(uint32_t c) { return c < 16 ? c : c < 20 ? 16 + 2 * (c - 16) : c < 22 ? 24 + 4 * (c - 20) : c < 24 ? 32 + 8 * (c - 22) : c == 24 ? 48 : 1u << (c - 19); }
__device__ __host__ inline uint32_t dec_ll_bits(uint32_t c) { return c < 16 ? 0 : c < 20 ? 1 : c < 22 ? 2 : c < 24 ? 3 : c == 24 ? 4 : c - 19; }
__device__ __host__ inline uint32_t dec_ml_base(uint32_t c) { return c < 32 ? c + 3 : c < 36 ? 35 + 2 * (c - 32) : c < 38 ? 43 + 4 * (c - 36) : c < 40 ? 51 + 8 * (c - 38) : c < 42 ? 67 + 16 * (c - 40) : c == 42 ? 99 : (1u << (c - 36)) + 3; }
__device__ __host__ inline uint32_t dec_ml_bits(uint32_t c) { return c < 32 ? 0 : c < 36 ? 1 : c < 38 ? 2 : c < 40 ? 3 : c < 42 ? 4 : c == 42 ? 5 : c - 36; }
__device__ __host__ inline uint32_t dec_of_base(uint32_t c) { return c == 0 ? 0 : c == 1 ? 1 : (1u << c) - 3; }
// default distributions (zstd_internal.h:150-190): LL log 6, OF log 5, ML log 6
__device__ __host__ inline int dec_default_norm(int kind, uint32_t s)
{
    if (kind == 0) return s == 0 ? 4 : s == 1 ? 3 : s <= 12 ? 2 : s <= 15 ? 1 : s <= 24 ? 2 : s == 25 ? 3 : s == 26 ? 2 : s <= 31 ? 1 : -1;
    if (kind == 1) return s <= 5 ? 1 : s <= 8 ? 2 : s <= 23 ? 1 : -1;
    return s == 0 ? 1 : s == 1 ? 4 : s == 2 ? 3 : s <= 8 ? 2 : s <= 45 ? 1 : -1;
}
__device__ __host__ inline uint32_t dec_max_sym(int kind) { return kind == 0 ? 35u : kind == 1 ? 31u : 52u; }
__device__ __host__ inline uint32_t dec_max_log(int kind) { return kind == 1 ? 8u : 9u; }
__device__ __host__ inline uint32_t dec_hb(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }

// FSE decoding entry = ZSTD_seqSymbol (zstd_decompress_block.h): nextState u16 | nbAdditionalBits u8 << 16 | nbBits u8 << 24 | baseValue << 32
__device__ __host__ inline uint64_t fse_d_pack(uint32_t next, uint32_t nbAdd, uint32_t nb, uint32_t base) { return (uint64_t)next | ((uint64_t)nbAdd << 16) | ((uint64_t)nb << 24) | ((uint64_t)base << 32); }
__device__ __host__ inline void dec_base_bits(int kind, uint32_t s, uint32_t* base, uint32_t* bits)
{
    if (kind == 0) { *base = dec_ll_base(s); *bits = dec_ll_bits(s); }
    else if (kind == 1) { *base = dec_of_base(s); *bits = s; }
    else { *base = dec_ml_base(s); *bits = dec_ml_bits(s); }
}

// ZSTD_buildFSETable_body (zstd_decompress_block.c:484-585), serial; T has 1 << tableLog entries, symOf/next are scratch.
// TT / TS / TN are pointer types so that the same code serves LDS (device) and plain memory (host-built dictionary tables).
template <typename TT, typename TS, typename TN, typename TNORM>
__device__ __host__ inline void fse_d_build(TT T, TS symOf, TN next, TNORM norm, uint32_t maxSym, int kind, uint32_t tableLog)
{
    uint32_t const tsz = 1u << tableLog, mask = tsz - 1, step = (tsz >> 1) + (tsz >> 3) + 3;
    uint32_t high = tsz - 1, pos = 0;
    for (uint32_t s = 0; s <= maxSym; s++) { int const c = norm[s]; if (c == -1) { symOf[high--] = (uint8_t)s; next[s] = 1; } else next[s] = (uint16_t)c; }
    for (uint32_t s = 0; s <= maxSym; s++) {
        int const c = norm[s];
        for (int i = 0; i < c; i++) { symOf[pos] = (uint8_t)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; }
    }
    for (uint32_t u = 0; u < tsz; u++) {
        uint32_t const sy = symOf[u]; uint32_t const ns = next[sy]; next[sy] = (uint16_t)(ns + 1);
        uint32_t const nb = tableLog - dec_hb(ns);
        uint32_t base, bits; dec_base_bits(kind, sy, &base, &bits);
        T[u] = fse_d_pack(((ns << nb) - tsz) & 0xFFFFu, bits, nb, base);
    }
}

// FSE_readNCount (lib/common/entropy_common.c:42-214) from a byte pointer readable up to `size`; returns bytes consumed or 0
template <typename TNORM>
__device__ __host__ inline uint32_t fse_d_read_ncount(TNORM norm, uint32_t* maxSymIO, uint32_t* tableLog, const uint8_t* src, uint32_t size)
{
    uint32_t const maxSV1 = *maxSymIO + 1;
    uint32_t bit = 0, charnum = 0; int remaining, threshold, nbBits; bool previous0 = false;
    auto peek = [&](uint32_t at, uint32_t n) -> uint32_t {
        uint64_t v = 0; uint32_t const byte = at >> 3;
        for (uint32_t i = 0; i < 5; i++) if (byte + i < size) v |= (uint64_t)src[byte + i] << (8 * i);
        return (uint32_t)((v >> (at & 7)) & ((1ull << n) - 1));
    };
    if (size == 0) return 0;
    for (uint32_t s = 0; s < maxSV1; s++) norm[s] = 0;
    nbBits = (int)peek(bit, 4) + 5; bit += 4;
    if (nbBits > 15) return 0;
    *tableLog = (uint32_t)nbBits;
    remaining = (1 << nbBits) + 1; threshold = 1 << nbBits; nbBits++;
    while (remaining > 1 && charnum < maxSV1) {
        if (previous0) {
            for (;;) { uint32_t const r = peek(bit, 2); bit += 2; charnum += r; if (r != 3) break; if (bit > 8 * size + 64) return 0; }
            if (charnum >= maxSV1) break;
        }
        int const mx = (2 * threshold - 1) - remaining;
        uint32_t const bits = peek(bit, (uint32_t)nbBits);
        int count;
        if ((int)(bits & (uint32_t)(threshold - 1)) < mx) { count = (int)(bits & (uint32_t)(threshold - 1)); bit += (uint32_t)nbBits - 1; }
        else { count = (int)(bits & (uint32_t)(2 * threshold - 1)); if (count >= threshold) count -= mx; bit += (uint32_t)nbBits; }
        count--;
        remaining -= count < 0 ? -count : count;
        norm[charnum++] = (int16_t)count;
        previous0 = !count;
        while (remaining < threshold) { nbBits--; threshold >>= 1; }
    }
    if (remaining != 1 || charnum > maxSV1 || bit > 8 * size) return 0;
    *maxSymIO = charnum - 1;
    return (bit + 7) >> 3;
}

// Double-symbol Huffman decoding entry for table index i (table log tl <= 11), from the single-symbol table T1 (symbol | nbBits << 8):
// the first code of the tl bits, and the second one when it also lies completely inside them (the idea of the reference's X2
// decoder, huf_decompress.c:953-1100, in a layout of our own): sym1 | sym2 << 8 | len1 << 16 | lenBoth << 20 | count << 28.
template <typename TP>
__device__ __host__ inline uint32_t huf_double_entry(TP T1, uint32_t tl, uint32_t i)
{
    uint32_t const e1 = T1[i], l1 = e1 >> 8;
    uint32_t const i2 = (i << l1) & ((1u << tl) - 1);          // the bits after the first code, zero-filled
    uint32_t const e2 = T1[i2], l2 = e2 >> 8;
    if (l1 + l2 <= tl) return (e1 & 0xFF) | ((e2 & 0xFF) << 8) | (l1 << 16) | ((l1 + l2) << 20) | (2u << 28);
    return (e1 & 0xFF) | (l1 << 16) | (l1 << 20) | (1u << 28);
}

#ifndef ZHIP_DECODE_HOST_ONLY
// optional phase profile (scripts/prof_decode.py, -DZHIP_PROF builds only): lane 0 of each wave adds s_memtime deltas to g_prof
#ifdef ZHIP_PROF
#define DPROF_BEGIN uint64_t dp_t_ = __builtin_amdgcn_s_memtime();
#define DPROF(slot) do { uint64_t const t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) atomicAdd(&zhip::g_prof[slot], (unsigned long long)(t_ - dp_t_)); dp_t_ = t_; } while (0)
#define DPROF_ADD(slot, v) do { if (lane == 0) atomicAdd(&zhip::g_prof[slot], (unsigned long long)(v)); } while (0)
#define DPROF_LOCAL uint64_t dl_t_ = __builtin_amdgcn_s_memtime(); uint64_t dl_a_[6] = {0, 0, 0, 0, 0, 0};
#define DPROF_L(i) do { uint64_t const t_ = __builtin_amdgcn_s_memtime(); dl_a_[i] += t_ - dl_t_; dl_t_ = t_; } while (0)
#define DPROF_LFLUSH(base) do { if (lane == 0) for (int i_ = 0; i_ < 6; i_++) atomicAdd(&zhip::g_prof[(base) + i_], (unsigned long long)dl_a_[i_]); } while (0)
#else
#define DPROF_LOCAL
#define DPROF_L(i) do { } while (0)
#define DPROF_LFLUSH(base) do { } while (0)
#define DPROF_BEGIN
#define DPROF(slot) do { } while (0)
#define DPROF_ADD(slot, v) do { } while (0)
#endif

// ------------------------------------------------------------------ LDS of one workgroup
struct DecShared {
    uint64_t fseAll[1280];           // LL[512] OF[256] ML[512] decoding tables (dec_tab())
    uint16_t huf[4096];              // symbol | nbBits << 8, 1 << hufLog entries
    int16_t  norm[3][64];
    uint16_t next[3][64];
    uint8_t  symOf[3][512];
    uint8_t  weights[256];
    uint16_t wNew[64]; uint8_t wSym[64], wNb[64];    // FSE table of the Huffman weights (table log <= 6)
    uint16_t hufStart[256];          // first table entry of each symbol while the Huffman table is being filled
    int16_t  wNorm[256]; uint16_t wNext[256];        // weights' FSE distribution while its table is built
    uint32_t ring[ZHIP_DEC_RING_WORDS + 4];         // the sequence bitstream, staged: word w = the 32 bits consumed w-th (seq_ring_fill);
                                                    // the last 4 words mirror the first 4 so that a window never wraps
    uint32_t bat[64][4];             // pass 1 -> pass 2: bit position and the three states of each sequence of a batch
    uint32_t frame;                  // queue ticket
    uint32_t status;                 // first error of the frame
    uint32_t hufLog, hufValid, fseValid;
    uint32_t dictHufIn, dictFseIn;   // the tables in LDS are still the dictionary's (survives from frame to frame of one launch)
    uint32_t log[3];
    uint32_t rep[3];
    uint32_t litSize, litMode, litByte, litSecSize;   // litMode 0: decoded into the literal buffer, 1: raw (in the source), 2: RLE
    uint32_t litSrcOff;              // raw literals: offset of the first literal in the block
    uint32_t nbSeq, seqDone;
    uint32_t cnt[2];                 // records in each hand-over buffer
    uint32_t endOut, endLit;         // positions after the last decoded sequence
};

__device__ __forceinline__ uint64_t* dec_tab(DecShared* S, uint32_t k) { return S->fseAll + (k == 0 ? 0u : k == 1 ? 512u : 768u); }

// ------------------------------------------------------------------ backward bit reader (lib/common/bitstream.h:250-420)
// Stream bytes [base, base+size); the last byte holds the end mark.  Bits are consumed from just below the mark downwards.
// `acc` holds upcoming bits at its top; refills come from `stash` (64 bits, loaded earlier) in 32-bit halves, and the next
// 8 bytes are requested the moment the stash is replaced, a whole 64 bits of decoding before they are needed.  Bytes
// before `base` read as zero; `used` against `total` is BIT_endOfDStream's test.  Every stream of a frame is preceded by at
// least 8 bytes of that frame (magic, descriptor, block header), so the 8-byte loads never leave the buffer.
struct BitsRev {
    const uint8_t* base; int32_t ptr;        // bytes [0, ptr) not requested yet
    uint64_t acc, stash, pend; int32_t n, half; int32_t loaded, total;
};
__device__ __forceinline__ uint64_t br_load(const uint8_t* base, int32_t& ptr)
{
    if (ptr >= 8) { ptr -= 8; return ld64(base + ptr); }
    if (ptr <= 0) return 0;
    uint64_t const v = ld64(base + ptr - 8) & (~0ull << (8 * (8 - ptr)));
    ptr = 0;
    return v;
}
// false = empty stream or missing end mark (bitstream.h:262, :284)
__device__ __forceinline__ bool br_init(BitsRev& b, const uint8_t* base, uint32_t size)
{
    b.base = base; b.ptr = (int32_t)size; b.n = 0; b.half = 0; b.loaded = 0; b.total = 0; b.acc = b.stash = b.pend = 0;
    if (size == 0) return false;
    uint32_t const last = base[size - 1];
    if (last == 0) return false;
    uint32_t const hb = dec_hb(last);
    uint64_t const first = br_load(base, b.ptr);            // the stream's top 8 bytes (zero-extended below its start)
    b.total = (int32_t)(8 * (size - 1) + hb);
    b.acc = (first << (7 - hb)) << 1;                       // drop the mark and what lies above it
    b.n = (int32_t)(56 + hb); b.loaded = b.n;
    b.stash = br_load(base, b.ptr);
    b.pend = br_load(base, b.ptr);
    return true;
}
__device__ __forceinline__ void br_refill(BitsRev& b)       // call when n <= 32; afterwards 32 < n <= 64
{
    b.acc |= (b.stash >> 32) << (32 - b.n);
    b.n += 32; b.loaded += 32;
    b.stash <<= 32;
    if (b.half) { b.stash = b.pend; b.pend = br_load(b.base, b.ptr); }
    b.half ^= 1;
}
__device__ __forceinline__ uint32_t br_read(BitsRev& b, uint32_t nb)     // nb <= 32, nb <= n
{
    uint32_t const v = nb ? (uint32_t)(b.acc >> (64 - nb)) : 0u;
    b.acc = nb >= 64 ? 0 : b.acc << nb; b.n -= (int32_t)nb;
    return v;
}
__device__ __forceinline__ int32_t br_used(const BitsRev& b) { return b.loaded - b.n; }

// ------------------------------------------------------------------ frame header (zstd_decompress.c:438-545)
struct DecHeader { uint32_t size; uint32_t checksum, single; uint32_t dictID; uint64_t fcs; uint32_t blockMax; uint32_t err; };
__device__ inline DecHeader dec_frame_header(const uint8_t* p, uint32_t n)
{
    DecHeader h; h.size = 0; h.checksum = 0; h.single = 0; h.dictID = 0; h.fcs = ~0ull; h.blockMax = ZHIP_UNIT_MAX; h.err = 0;
    if (n < 5 + 3) { h.err = ZHIP_DE_SRC_WRONG; return h; }
    if (ld32(p) != 0xFD2FB528u) { h.err = ZHIP_DE_PREFIX; return h; }
    uint32_t const fhd = p[4], didCode = fhd & 3, fcsCode = fhd >> 6;
    h.checksum = (fhd >> 2) & 1; h.single = (fhd >> 5) & 1;
    if (fhd & 8) { h.err = ZHIP_DE_UNSUPPORTED; return h; }
    uint32_t const didB = didCode == 3 ? 4 : didCode, fcsB = fcsCode == 0 ? (h.single ? 1u : 0u) : (1u << fcsCode);
    h.size = 5 + (h.single ? 0u : 1u) + didB + fcsB;
    if (n < h.size + 3) { h.err = ZHIP_DE_SRC_WRONG; return h; }
    uint32_t pos = 5; uint64_t window = 0;
    if (!h.single) {
        uint32_t const wl = (p[pos] >> 3) + 10;
        if (wl > 31) { h.err = ZHIP_DE_WINDOW; return h; }
        window = 1ull << wl; window += (window >> 3) * (p[pos] & 7); pos++;
    }
    for (uint32_t i = 0; i < didB; i++) h.dictID |= (uint32_t)p[pos + i] << (8 * i);
    pos += didB;
    if (fcsB) { uint64_t v = 0; for (uint32_t i = 0; i < fcsB; i++) v |= (uint64_t)p[pos + i] << (8 * i); if (fcsCode == 1) v += 256; h.fcs = v; }
    if (h.single) window = h.fcs;
    h.blockMax = window < ZHIP_UNIT_MAX ? (uint32_t)window : ZHIP_UNIT_MAX;
    return h;
}

// ------------------------------------------------------------------ literals section, wave 0
// header fields (zstd_decompress_block.c:134-345), computed by every lane that needs them
struct LitHeader { uint32_t type, lh, litSize, cSize, single, err; };
__device__ inline LitHeader dec_lit_header(const uint8_t* ip, uint32_t size, uint32_t blockMax)
{
    LitHeader h; h.err = 0; h.single = 0; h.cSize = 0; h.lh = 0; h.litSize = 0; h.type = 0;
    if (size < 2) { h.err = ZHIP_DE_CORRUPT; return h; }
    uint32_t const b0 = ip[0], sf = (b0 >> 2) & 3;
    h.type = b0 & 3;
    if (h.type >= 2) {
        if (size < 5) { h.err = ZHIP_DE_CORRUPT; return h; }
        uint32_t const lhc = ld32(ip);
        if (sf < 2) { h.single = !sf; h.lh = 3; h.litSize = (lhc >> 4) & 0x3FF; h.cSize = (lhc >> 14) & 0x3FF; }
        else if (sf == 2) { h.lh = 4; h.litSize = (lhc >> 4) & 0x3FFF; h.cSize = lhc >> 18; }
        else { h.lh = 5; h.litSize = (lhc >> 4) & 0x3FFFF; h.cSize = (lhc >> 22) + ((uint32_t)ip[4] << 10); }
        if (h.litSize > blockMax || (!h.single && h.litSize < 6) || h.cSize + h.lh > size) h.err = ZHIP_DE_CORRUPT;
    } else {
        if (sf == 0 || sf == 2) { h.lh = 1; h.litSize = b0 >> 3; }
        else if (sf == 1) { h.lh = 2; h.litSize = (b0 | ((uint32_t)ip[1] << 8)) >> 4; }               // byte loads: the block may end right here
        else { h.lh = 3; if (size < 3) { h.err = ZHIP_DE_CORRUPT; return h; } h.litSize = (b0 | ((uint32_t)ip[1] << 8) | ((uint32_t)ip[2] << 16)) >> 4; }
        h.cSize = h.type == 0 ? h.litSize : 1;
        if (h.litSize > blockMax || h.lh + h.cSize > size) h.err = ZHIP_DE_CORRUPT;
    }
    return h;
}

// Huffman weights compressed with FSE (lib/common/fse_decompress.c:58-277): lane 0 only. returns the number of weights, 0 = error
__device__ inline uint32_t dec_fse_weights(DecShared* S, const uint8_t* src, uint32_t size)
{
    uint32_t maxSym = 255, tl;
    int16_t* const norm = S->wNorm;
    uint32_t const h = fse_d_read_ncount(norm, &maxSym, &tl, src, size);
    if (!h || tl > 6 || h >= size) return 0;
    {   uint32_t const tsz = 1u << tl, mask = tsz - 1, step = (tsz >> 1) + (tsz >> 3) + 3; uint32_t high = tsz - 1, pos = 0;
        uint16_t* const nx = S->wNext;
        for (uint32_t s = 0; s <= maxSym; s++) { if (norm[s] == -1) { S->wSym[high--] = (uint8_t)s; nx[s] = 1; } else nx[s] = (uint16_t)norm[s]; }
        for (uint32_t s = 0; s <= maxSym; s++) for (int i = 0; i < norm[s]; i++) { S->wSym[pos] = (uint8_t)s; pos = (pos + step) & mask; while (pos > high) pos = (pos + step) & mask; }
        if (pos != 0) return 0;
        for (uint32_t u = 0; u < tsz; u++) { uint32_t const ns = nx[S->wSym[u]]++; uint32_t const nb = tl - dec_hb(ns); S->wNb[u] = (uint8_t)nb; S->wNew[u] = (uint16_t)((ns << nb) - tsz); }
    }
    BitsRev b;
    if (!br_init(b, src + h, size - h)) return 0;
    uint32_t s1 = br_read(b, tl), s2 = br_read(b, tl), n = 0; bool which = false;
    for (;;) {                                                 // fse_decompress.c:207-233
        if (b.n <= 32) br_refill(b);
        uint32_t const st = which ? s2 : s1, other = which ? s1 : s2;
        uint32_t const nb = S->wNb[st];
        if (n + 2 > 255) return 0;
        S->weights[n++] = S->wSym[st];
        uint32_t const ns = S->wNew[st] + br_read(b, nb);
        if (which) s2 = ns; else s1 = ns;
        if (br_used(b) > b.total) { S->weights[n++] = S->wSym[other]; break; }
        which = !which;
    }
    return n;
}

// tree description -> decoding table in LDS (HUF_readStats + HUF_readDTableX1_wksp); whole wave 0, control on lane 0.
// returns bytes consumed, 0 = error
// (always inlined: since the block-parallel decoder of zhip_decode_big.h calls it too the compiler would otherwise make it a real function — a different k_decode)
__device__ __attribute__((always_inline)) inline uint32_t dec_huf_table(DecShared* S, const uint8_t* src, uint32_t size)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t ok = 0, consumed = 0, nbSym = 0, tableLog = 0;
    if (lane == 0 && size) {
        uint32_t iSize = src[0], oSize = 0; bool good = true;
        if (iSize >= 128) {
            oSize = iSize - 127; iSize = (oSize + 1) / 2;
            if (iSize + 1 > size) good = false;
            else for (uint32_t n = 0; n < oSize; n += 2) { S->weights[n] = src[1 + n / 2] >> 4; S->weights[n + 1] = src[1 + n / 2] & 15; }
        } else {
            if (iSize + 1 > size) good = false;
            else { oSize = dec_fse_weights(S, src + 1, iSize); if (!oSize) good = false; }
        }
        if (good) {
            uint32_t rank[16], total = 0;
            for (int i = 0; i < 16; i++) rank[i] = 0;
            for (uint32_t n = 0; n < oSize; n++) { uint32_t const w = S->weights[n]; if (w > 12) { good = false; break; } rank[w]++; total += (1u << w) >> 1; }
            if (good && total) {
                tableLog = dec_hb(total) + 1;
                uint32_t const rest = (1u << tableLog) - total;
                if (tableLog > 12 || (1u << dec_hb(rest)) != rest) good = false;
                else {
                    uint32_t const last = dec_hb(rest) + 1;
                    S->weights[oSize] = (uint8_t)last; rank[last]++;
                    if (rank[1] < 2 || (rank[1] & 1)) good = false;
                }
            } else good = false;
            if (good) {
                // table ranges: weight 1 symbols first, then weight 2, ... each in symbol order (huf_decompress.c:427-480)
                uint32_t start[14], pos = 0;
                for (uint32_t w = 1; w <= tableLog; w++) { start[w] = pos; pos += rank[w] << (w - 1); }
                nbSym = oSize + 1;
                for (uint32_t n = 0; n < nbSym; n++) {
                    uint32_t const w = S->weights[n];
                    uint32_t st = 0;
                    if (w) { st = start[w]; start[w] += (1u << w) >> 1; }
                    S->hufStart[n] = (uint16_t)st;
                }
                ok = 1; consumed = iSize + 1;
            }
        }
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    if (!ok) return 0;
    consumed = __builtin_amdgcn_readfirstlane(consumed); nbSym = __builtin_amdgcn_readfirstlane(nbSym); tableLog = __builtin_amdgcn_readfirstlane(tableLog);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t n = 0; n < nbSym; n++) {                     // wave-uniform loop; lanes fill the symbol's range
        uint32_t const w = S->weights[n];
        if (!w) continue;
        uint32_t const len = (1u << w) >> 1, st = S->hufStart[n];
        uint16_t const e = (uint16_t)(n | ((tableLog + 1 - w) << 8));
        for (uint32_t k = lane; k < len; k += 64) S->huf[st + k] = e;
    }
    __builtin_amdgcn_wave_barrier();
    if (tableLog <= 11) {                                       // double-symbol form, in place: every lane computes its entries from the
        uint32_t regs[32]; uint32_t const tsz = 1u << tableLog;  // single-symbol table first, then they are written over it
#pragma unroll
        for (int r = 0; r < 32; r++) { uint32_t const i = (uint32_t)r * 64 + lane; regs[r] = i < tsz ? huf_double_entry(S->huf, tableLog, i) : 0; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 32; r++) { uint32_t const i = (uint32_t)r * 64 + lane; if (i < tsz) ((uint32_t*)S->huf)[i] = regs[r]; }
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) { S->hufLog = tableLog; S->hufValid = 1; S->dictHufIn = 0; }
    __builtin_amdgcn_wave_barrier();
    return consumed;
}

// ---- the four (or one) Huffman streams -> the literal buffer (huf_decompress.c:560-700), decoded by ALL lanes: 16 lanes per stream, each owning a contiguous range of the stream's bits.
// A prefix code resynchronises: two decoders started at different bit positions fall onto the same codeword boundaries
// after a few symbols.  So (1) every lane decodes a short run-in (ZHIP_HUF_RUNIN bits above its range) from a guessed
// position to find where the true chain enters its successor's range, (2) every lane decodes its own range from the entry
// its predecessor reported, counting symbols and reporting its exit — repeated until no entry changes (the first lane's
// entry is the stream's top, so a fixed point IS the serial decode), (3) a prefix sum of the counts gives each lane its
// output offset and it decodes once more, storing symbols.  Exactness: same symbols as the serial decoder; the stream is
// accepted iff the chain ends on bit 0 with exactly the expected number of symbols (BIT_endOfDStream + op == oend).
#ifndef ZHIP_HUF_RUNIN
#define ZHIP_HUF_RUNIN 192
#endif
// Bit reader of the parallel decoder.  All lanes reload at the SAME moments (every four symbols), so the wave never runs a
// refill path for the sake of one lane, and the bytes of reload k+1 are requested at reload k: the next read point is at most
// 7 bytes below the current one (5 lookups x <= 11 bits, or 4 x 12), so a 16-byte window that starts 8 bytes below the current chunk
// always contains the next chunk.  pos = bits of the stream below the read point; bits below the stream's start read as the
// bytes that precede it (never consumed by a valid stream: its chain ends exactly on bit 0, anything else is rejected).
struct BitsAt { const uint8_t* base; int32_t pos; int32_t wa; uint64_t wlo, whi; int32_t okLo, okHi; };   // window = 16 bytes at byte offset wa
__device__ __forceinline__ void ba_window(BitsAt& b, int32_t chunkAt)
{
    int32_t a = chunkAt - 8; if (a < -8) a = -8;
    uint64_t v[2];
    if (a >= b.okLo && a + 16 <= b.okHi) __builtin_memcpy(v, b.base + a, 16);
    else {                                                      // a window that would leave the frame (tiny streams): byte by byte
        v[0] = v[1] = 0;
        for (int i = 0; i < 16; i++) { int32_t const q = a + i; if (q >= b.okLo && q < b.okHi) v[i >> 3] |= (uint64_t)b.base[q] << (8 * (i & 7)); }
    }
    b.wa = a; b.wlo = v[0]; b.whi = v[1];
}
// [okLo, okHi) = byte offsets around base that are known to lie inside the frame
__device__ __forceinline__ void ba_init(BitsAt& b, const uint8_t* base, int32_t pos, int32_t okLo, int32_t okHi)
{
    b.base = base; b.pos = pos; b.okLo = okLo; b.okHi = okHi; ba_window(b, ((pos + 7) >> 3) - 8);
}
// the 8-byte chunk that ends at the byte holding bit pos-1, left-aligned on that bit; requests the next window
__device__ __forceinline__ uint64_t ba_chunk(BitsAt& b)
{
    int32_t const top = (b.pos + 7) >> 3, at = top - 8;        // chunk = bytes [at, at + 8)
    int32_t off = at - b.wa; if (off < 0) off = 0;             // 0..8 (below 0 only once the stream is exhausted)
    uint32_t const sh = 8u * (uint32_t)off;
    uint64_t const chunk = sh == 0 ? b.wlo : (sh >= 64 ? b.whi : ((b.wlo >> sh) | (b.whi << (64 - sh))));
    ba_window(b, at);
    return chunk << ((uint32_t)(8 * top - b.pos) & 63);
}
// decode from b.pos down to (and possibly past) `lo`; returns the number of symbols, leaves the exit position in b.pos
__device__ __forceinline__ uint32_t huf_run(BitsAt& b, const lds_u16* T, uint32_t sh, int32_t lo)
{
    uint32_t cnt = 0;
    while (b.pos > lo) {
        uint64_t acc = ba_chunk(b);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t const e = T[(uint32_t)(acc >> sh)];
            bool const live = b.pos > lo;
            acc <<= (e >> 8);
            if (live) { b.pos -= (int32_t)(e >> 8); cnt++; }
        }
    }
    return cnt;
}
// the same with the double-symbol table (table log <= 11): up to two symbols per lookup; the second one is taken only when it
// starts above `lo`, so the exit position is the same codeword boundary the single-symbol walk finds
__device__ __forceinline__ uint32_t huf_run2(BitsAt& b, const lds_u32* T2, uint32_t sh, int32_t lo)
{
    uint32_t cnt = 0;
    // far from the range's end nothing needs checking: five lookups take at most 55 of the chunk's >= 57 bits, every one of them
    // is inside the range and both of its symbols count (the entry's total length and count fields are used as they are)
    while (b.pos - 55 > lo) {
        uint64_t acc = ba_chunk(b);
#pragma unroll
        for (int k = 0; k < 5; k++) {
            uint32_t const e = T2[(uint32_t)(acc >> sh)];
            uint32_t const lt = (e >> 20) & 31;
            acc <<= lt; b.pos -= (int32_t)lt; cnt += e >> 28;
        }
    }
    while (b.pos > lo) {
        uint64_t acc = ba_chunk(b);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t const e = T2[(uint32_t)(acc >> sh)];
            int32_t const l1 = (int32_t)((e >> 16) & 15), lt = (int32_t)((e >> 20) & 31);
            bool const live = b.pos > lo, two = (e >> 28) == 2 && b.pos - l1 > lo;
            int32_t const used = two ? lt : l1;
            acc <<= (uint32_t)used;
            if (live) { b.pos -= used; cnt += two ? 2u : 1u; }
        }
    }
    return cnt;
}
__device__ __attribute__((always_inline)) inline uint32_t dec_huf_streams_par(DecShared* S, const uint8_t* src, uint32_t size, uint32_t litSize, bool single, uint8_t* lit)
{
    uint32_t const lane = (uint32_t)lane_id(), grp = lane >> 4, j = lane & 15;
    uint32_t const tl = S->hufLog, sh = 64 - tl;
    const lds_u16* const T = (const lds_u16*)(uintptr_t)S->huf;
    const lds_u32* const T2 = (const lds_u32*)(uintptr_t)S->huf;
    bool const dbl = tl <= 11;                                 // which form the table in LDS has (dec_huf_table / dictionary copy)
    uint32_t sOff = 0, sLen = size, oOff = 0, oLen = litSize;
    bool mine = grp == 0;                                      // this lane's stream exists
    if (!single) {
        if (size < 10) return ZHIP_DE_CORRUPT;
        uint32_t const l1 = src[0] | (src[1] << 8), l2 = src[2] | (src[3] << 8), l3 = src[4] | (src[5] << 8);
        if (6 + l1 + l2 + l3 > size) return ZHIP_DE_CORRUPT;
        uint32_t const seg = (litSize + 3) / 4;
        if (3 * seg > litSize) return ZHIP_DE_CORRUPT;
        mine = true;
        sOff = 6 + (grp >= 1 ? l1 : 0) + (grp >= 2 ? l2 : 0) + (grp >= 3 ? l3 : 0);
        sLen = grp == 0 ? l1 : grp == 1 ? l2 : grp == 2 ? l3 : size - 6 - l1 - l2 - l3;
        oOff = seg * grp; oLen = grp < 3 ? seg : litSize - 3 * seg;
    }
    const uint8_t* const base = src + sOff;
    int32_t const okLo = -(int32_t)sOff - 8, okHi = (int32_t)size - (int32_t)sOff;     // the literal payload and the 8 bytes of headers before it
    bool bad = false; int32_t B = 0;
    if (mine) {
        uint32_t const last = sLen ? base[sLen - 1] : 0;
        if (!last) bad = true; else B = (int32_t)(8 * (sLen - 1) + dec_hb(last));
    }
    if (__any(bad)) return ZHIP_DE_CORRUPT;
    int32_t const per = ((B + 15) >> 4) < 64 ? 64 : ((B + 15) >> 4);          // bits per lane (short streams use fewer lanes)
    int32_t const hi = B - (int32_t)j * per, lo = hi - per > 0 ? hi - per : 0;
    bool const on = mine && (hi > 0 || (j == 0));              // lane 0 of a stream is always on (B may be 0: no data bits)
    // (1) run-in: where does a chain started a little above my range leave it?  (guess for my successor's entry)
    int32_t exitPos = hi;
    if (on && hi > 0) {
        if (j == 0) exitPos = B;                                // the true start; decoded in step (2)
        else { int32_t const st = hi + ZHIP_HUF_RUNIN < B ? hi + ZHIP_HUF_RUNIN : B; BitsAt b; ba_init(b, base, st, okLo, okHi); if (dbl) (void)huf_run2(b, T2, sh, hi); else (void)huf_run(b, T, sh, hi); exitPos = b.pos; }
    }
    // after the run-in lane j holds a guess for ITS OWN entry (the chain's first position <= hi); lane 0's is exact
    int32_t entry = exitPos;
    uint32_t cnt = 0; int32_t myExit = entry;
    bool dirty = on;
    for (int round = 0; round < 20; round++) {
        if (dirty) {
            if (entry > lo) { BitsAt b; ba_init(b, base, entry, okLo, okHi); cnt = dbl ? huf_run2(b, T2, sh, lo) : huf_run(b, T, sh, lo); myExit = b.pos; }
            else { cnt = 0; myExit = entry; }
        }
        int32_t const predExit = __shfl(myExit, (int)(lane - 1));
        bool const changed = on && j > 0 && predExit != entry;
        if (changed) entry = predExit;
        dirty = changed;
        if (!__any(changed)) break;
        if (round == 19) return ZHIP_DE_CORRUPT;                // never observed: chains that refuse to merge for 20 rounds
    }
    // validity: the chain must end exactly on bit 0 and hold exactly oLen symbols
    uint32_t pre = 0, tot = 0;
    {   uint32_t c = on ? cnt : 0;
        for (int k = 0; k < 16; k++) { uint32_t const ck = __shfl(c, (int)((lane & ~15u) + (uint32_t)k)); if ((uint32_t)k < j) pre += ck; tot += ck; }
    }
    bool const isLast = on && lo == 0;                          // the lane whose range reaches the stream's start
    if (mine && tot != oLen) bad = true;
    if (isLast && myExit != 0) bad = true;
    if (__any(bad)) return ZHIP_DE_CORRUPT;
    // (3) output
    if (on && cnt && dbl) {
        BitsAt b; ba_init(b, base, entry, okLo, okHi);
        uint8_t* const o = lit + oOff + pre; uint32_t i = 0;
        // while at least 10 symbols remain, five lookups (<= 10 symbols) cannot overrun the lane's region: each stores two bytes, a
        // single-symbol entry's second byte is overwritten by the next store
        while (i + 10 <= cnt) {
            uint64_t acc = ba_chunk(b);
#pragma unroll
            for (int k = 0; k < 5; k++) {
                uint32_t const e = T2[(uint32_t)(acc >> sh)];
                uint32_t const lt = (e >> 20) & 31;
                acc <<= lt; b.pos -= (int32_t)lt;
                uint16_t const w = (uint16_t)e; __builtin_memcpy(o + i, &w, 2);
                i += e >> 28;
            }
        }
        while (i < cnt) {
            uint64_t acc = ba_chunk(b);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t const e = T2[(uint32_t)(acc >> sh)];
                bool const live = i < cnt, two = (e >> 28) == 2 && i + 1 < cnt;
                uint32_t const used = two ? (e >> 20) & 31 : (e >> 16) & 15;
                acc <<= used; b.pos -= (int32_t)used;
                if (live) {
                    if (two) { uint16_t const w = (uint16_t)e; __builtin_memcpy(o + i, &w, 2); i += 2; }
                    else { o[i] = (uint8_t)e; i += 1; }
                }
            }
        }
    } else if (on && cnt) {
        BitsAt b; ba_init(b, base, entry, okLo, okHi);
        uint8_t* const o = lit + oOff + pre; uint32_t i = 0;
        while (i + 4 <= cnt) {
            uint64_t acc = ba_chunk(b);
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t const e = T[(uint32_t)(acc >> sh)];
                acc <<= (e >> 8); b.pos -= (int32_t)(e >> 8);
                w |= (e & 0xFF) << (8 * k);
            }
            __builtin_memcpy(o + i, &w, 4);
            i += 4;
        }
        if (i < cnt) {
            uint64_t acc = ba_chunk(b);
            for (; i < cnt; i++) { uint32_t const e = T[(uint32_t)(acc >> sh)]; acc <<= (e >> 8); o[i] = (uint8_t)e; }
        }
    }
    return 0;
}

// literals section of one block: wave 0.  Publishes litMode / litSize / litByte / litSrcOff in S (lane 0).
__device__ inline void dec_literals(DecShared* S, const uint8_t* blk, uint32_t bsize, uint32_t blockMax, uint8_t* lit, uint32_t dstRoom)
{
    uint32_t const lane = (uint32_t)lane_id();
    LitHeader const h = dec_lit_header(blk, bsize, blockMax);
    uint32_t err = h.err;
    if (!err && h.litSize > dstRoom) err = ZHIP_DE_DST_SMALL;
    uint32_t mode = 0;
    if (!err) {
        if (h.type >= 2) {
            const uint8_t* hs = blk + h.lh; uint32_t hn = h.cSize;
            if (h.type == 3) { if (!S->hufValid) err = ZHIP_DE_DICT_CORRUPT; }
            else {
                uint32_t const t = dec_huf_table(S, hs, hn);
                if (!t || t >= hn) err = ZHIP_DE_CORRUPT; else { hs += t; hn -= t; }
            }
            if (!err) err = dec_huf_streams_par(S, hs, hn, h.litSize, h.single != 0, lit);
        } else mode = h.type == 0 ? 1 : 2;
    }
    if (lane == 0) {
        if (err) atomicMax(&S->status, err);
        S->litMode = mode; S->litSize = h.litSize; S->litSecSize = h.lh + h.cSize;
        S->litSrcOff = h.lh; S->litByte = mode == 2 && !err ? blk[h.lh] : 0;
    }
}

// ------------------------------------------------------------------ sequences section, wave 1
// The interleaved bitstream is consumed from its last byte downwards.  D = "down position": the number of bits consumed so
// far counting from the top bit of the last byte; word w of the ring holds the bits D in [32w, 32w+32), most significant
// first, which is simply the little-endian dword that ends 4w bytes below the stream's end.  Any field can then be
// extracted at any D with two aligned LDS reads — no sequential container, so the three next-state fields of a sequence are
// fetched at once and the value bits can be fetched by other lanes later.
struct SeqDec {                     // wave 1's registers across hand-over chunks; every field is wave-uniform
    const uint8_t* base; uint32_t size;     // the bitstream
    uint32_t Dpos;                  // down position of the next sequence
    uint32_t wLoaded;               // ring holds words [wLoaded - ZHIP_DEC_RING_WORDS, wLoaded)
    uint32_t sLL, sOF, sML; uint32_t rep0, rep1, rep2; uint32_t outPos, litPos; uint32_t done;
};
// load ring words [w0, w0 + count) (count a multiple of 64), whole wave
__device__ __forceinline__ void seq_ring_fill(DecShared* S, const SeqDec& D, uint32_t w0, uint32_t count)
{
    uint32_t const lane = (uint32_t)lane_id();
    for (uint32_t w = w0 + lane; w < w0 + count; w += 64) {
        int32_t const at = (int32_t)D.size - 4 * (int32_t)(w + 1);              // byte offset of the dword in the stream
        uint32_t v = 0;
        if (at > -4) {                                                           // at least one byte inside the stream; bytes before
            v = ld32(D.base + at);                                               // its start belong to the same frame (>= 8 of them)
            if (at < 0) v &= ~0u << (8 * (uint32_t)(-at));
        }
        uint32_t const ri = w & (ZHIP_DEC_RING_WORDS - 1);
        S->ring[ri] = v;
        if (ri < 4) S->ring[ZHIP_DEC_RING_WORDS + ri] = v;
    }
}
// nb <= 32 bits at down position d
__device__ __forceinline__ uint32_t seq_field(const lds_u32* R, uint32_t d, uint32_t nb)
{
    uint32_t const w = (d >> 5) & (ZHIP_DEC_RING_WORDS - 1);
    uint64_t const x = ((uint64_t)R[w] << 32) | R[w + 1];
    return (uint32_t)(((x << (d & 31)) >> 32) >> (32 - nb));
}

// ZSTD_buildFSETable_body (zstd_decompress_block.c:484-585) by the whole wavefront, same table as the serial fse_d_build:
//   * symbols (<= 53, one lane each): low-probability symbols take the top cells, a wave scan gives the others their first rank;
//   * cells in the reference's visiting order i -> (i * step) & mask, 64 at a time: the visit's rank among the cells below the
//     low-probability area (ballot + popcount) names its symbol by a binary search in the ranks;
//   * cells in table order, 64 at a time: a cell's state number is its symbol's count plus the symbol's cells before it — lanes of
//     one symbol are grouped with ballots, the running count per symbol lives in LDS.
__device__ inline void fse_d_build_wave(DecShared* S, uint32_t k, uint32_t maxSym, uint32_t tableLog)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const tsz = 1u << tableLog, mask = tsz - 1, step = (tsz >> 1) + (tsz >> 3) + 3;
    uint64_t* const T = dec_tab(S, k);
    uint8_t* const symOf = S->symOf[k]; uint16_t* const nextArr = S->next[k];
    uint32_t* const cum = &S->bat[0][0];                       // 64 exclusive ranks (the hand-over area is free during table builds)
    int const c = lane <= maxSym ? (int)S->norm[k][lane] : 0;
    bool const low = c == -1;
    unsigned long long const lowMask = __ballot(low);
    uint32_t const nLow = (uint32_t)__popcll(lowMask), high = tsz - 1 - nLow;
    if (low) symOf[tsz - 1 - (uint32_t)__popcll(lowMask & below_mask((int)lane))] = (uint8_t)lane;
    uint32_t const cnt = c > 0 ? (uint32_t)c : 0;
    uint32_t inc = cnt;
    for (int sft = 1; sft < 64; sft <<= 1) { uint32_t const a = __shfl_up(inc, (unsigned)sft); if ((int)lane >= sft) inc += a; }
    cum[lane] = inc - cnt;
    if (lane <= maxSym) nextArr[lane] = (uint16_t)(low ? 1u : cnt);
    __builtin_amdgcn_wave_barrier();
    uint32_t rankBase = 0;
    for (uint32_t i0 = 0; i0 < tsz; i0 += 64) {
        uint32_t const i = i0 + lane, p = (i * step) & mask;
        bool const ok = i < tsz && p <= high;
        unsigned long long const okMask = __ballot(ok);
        uint32_t const rank = rankBase + (uint32_t)__popcll(okMask & below_mask((int)lane));
        rankBase += (uint32_t)__popcll(okMask);
        if (ok) {                                               // last symbol whose first rank is <= rank
            uint32_t lo = 0, hi = 63;
            while (lo < hi) { uint32_t const mid = (lo + hi + 1) >> 1; if (cum[mid] <= rank) lo = mid; else hi = mid - 1; }
            symOf[p] = (uint8_t)lo;
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t u0 = 0; u0 < tsz; u0 += 64) {
        uint32_t const u = u0 + lane; bool const on = u < tsz;
        uint32_t const sy = on ? symOf[u] : 0xFFu;
        uint32_t ns = 0;
        unsigned long long rest = __ballot(on);
        while (rest) {
            uint32_t const sL = __builtin_amdgcn_readlane(sy, first_lane(rest));
            unsigned long long const m = __ballot(on && sy == sL);
            uint32_t const base = nextArr[sL];
            if (on && sy == sL) ns = base + (uint32_t)__popcll(m & below_mask((int)lane));
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) nextArr[sL] = (uint16_t)(base + (uint32_t)__popcll(m));
            __builtin_amdgcn_wave_barrier();
            rest &= ~m;
        }
        if (on) {
            uint32_t const nb = tableLog - dec_hb(ns);
            uint32_t bs, bits; dec_base_bits((int)k, sy, &bs, &bits);
            T[u] = fse_d_pack(((ns << nb) - tsz) & 0xFFFFu, bits, nb, bs);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// header + the three tables (ZSTD_decodeSeqHeaders :662-745, ZSTD_buildSeqTable :625-660).  seq = the sequences section.
// Publishes nbSeq; on success initialises D.  Wave-uniform result: error code or 0.
__device__ inline uint32_t dec_seq_setup(DecShared* S, const uint8_t* seq, uint32_t size, SeqDec& D, const uint64_t* defTabs, uint32_t* nbSeqOut)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t err = 0, nbSeq = 0, pos = 0, modes = 0;
    uint32_t hdr[3] = {0, 0, 0};            // per table: 0 = nothing to build, 1 = build from S->norm, 2 = RLE symbol in rle[], 3 = copy the predefined table
    uint32_t rle[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, lg[3] = {0, 0, 0};
    if (lane == 0) {
        if (size < 1) err = ZHIP_DE_SRC_WRONG;
        else {
            nbSeq = seq[pos++];
            if (nbSeq > 0x7F) {
                if (nbSeq == 0xFF) { if (pos + 2 > size) err = ZHIP_DE_SRC_WRONG; else { nbSeq = (seq[pos] | (seq[pos + 1] << 8)) + 0x7F00; pos += 2; } }
                else { if (pos >= size) err = ZHIP_DE_SRC_WRONG; else { nbSeq = ((nbSeq - 0x80) << 8) + seq[pos++]; } }
            }
        }
        if (!err && nbSeq == 0 && pos != size) err = ZHIP_DE_CORRUPT;
        if (!err && nbSeq) {
            if (pos + 1 > size) err = ZHIP_DE_SRC_WRONG;
            else {
                modes = seq[pos++];
                if (modes & 3) err = ZHIP_DE_CORRUPT;
                for (int k = 0; k < 3 && !err; k++) {
                    uint32_t const type = (modes >> (6 - 2 * k)) & 3;
                    if (type == 0) { hdr[k] = 3; lg[k] = k == 1 ? 5 : 6; }
                    else if (type == 1) {
                        if (pos >= size || seq[pos] > dec_max_sym(k)) err = ZHIP_DE_CORRUPT;
                        else { hdr[k] = 2; rle[k] = seq[pos++]; lg[k] = 0; }
                    } else if (type == 3) { if (!S->fseValid) err = ZHIP_DE_CORRUPT; else lg[k] = S->log[k]; }
                    else {
                        uint32_t m = dec_max_sym(k), tl = 0;
                        uint32_t const h = fse_d_read_ncount(S->norm[k], &m, &tl, seq + pos, size - pos);
                        if (!h || tl > dec_max_log(k)) err = ZHIP_DE_CORRUPT;
                        else { hdr[k] = 1; mx[k] = m; lg[k] = tl; pos += h; }
                    }
                }
            }
        }
    }
    err = __builtin_amdgcn_readfirstlane(err);
    nbSeq = __builtin_amdgcn_readfirstlane(nbSeq);
    if (lane == 0) { S->nbSeq = err ? 0 : nbSeq; }
    *nbSeqOut = err ? 0 : nbSeq;                                // wave-uniform copy for the calling wave (S->nbSeq is for the other one, after the barrier)
    if (err) return err;
    if (!nbSeq) return 0;
    pos = __builtin_amdgcn_readfirstlane(pos);
    // build, table by table, the whole wave on each (fse_d_build_wave); predefined tables are copied from the constant
    for (int k = 0; k < 3; k++) {
        uint32_t const a = __builtin_amdgcn_readfirstlane(hdr[k]), r = __builtin_amdgcn_readfirstlane(rle[k]);
        uint32_t const m = __builtin_amdgcn_readfirstlane(mx[k]), l = __builtin_amdgcn_readfirstlane(lg[k]);
        if (a == 3) {                                          // predefined table: 64 / 32 entries
            uint32_t const n = 1u << l; const uint64_t* const src = defTabs + (k == 0 ? 0 : k == 1 ? 64 : 96);
            for (uint32_t i = lane; i < n; i += 64) dec_tab(S, (uint32_t)k)[i] = src[i];
        } else if (a == 1) fse_d_build_wave(S, (uint32_t)k, m, l);
        else if (a == 2) { if (lane == 0) { uint32_t base, bits; dec_base_bits(k, r, &base, &bits); dec_tab(S, (uint32_t)k)[0] = fse_d_pack(0, bits, 0, base); } }
        if (lane == 0) { S->log[k] = l; if (a) S->dictFseIn = 0; }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) S->fseValid = 1;
    // bitstream + initial states (zstd_decompress_block.c:1636-1643)
    uint32_t const bsz = size - pos;
    if (bsz == 0) return ZHIP_DE_CORRUPT;
    uint32_t const lastByte = seq[size - 1];
    if (lastByte == 0) return ZHIP_DE_CORRUPT;                  // no end mark (bitstream.h:284)
    D.base = seq + pos; D.size = bsz;
    D.wLoaded = ZHIP_DEC_RING_WORDS;
    {   // short streams (small frames) need only their own words; what lies beyond the stream's start reads as zero anyway, but
        // the first batch may look 3 words past its last bit, so round up generously to whole 64-word steps
        uint32_t const need = (((bsz + 3) >> 2) + 4 + 63) & ~63u;
        seq_ring_fill(S, D, 0, need < ZHIP_DEC_RING_WORDS ? need : ZHIP_DEC_RING_WORDS);
    }
    __builtin_amdgcn_wave_barrier();
    {   const lds_u32* const R = (const lds_u32*)(uintptr_t)S->ring;
        uint32_t const l0 = S->log[0], l1 = S->log[1], l2 = S->log[2];
        uint32_t d = 8 - dec_hb(lastByte);                      // the mark and the zero bits above it
        D.sLL = seq_field(R, d, l0); d += l0;
        D.sOF = seq_field(R, d, l1); d += l1;
        D.sML = seq_field(R, d, l2); d += l2;
        D.Dpos = d;
    }
    D.rep0 = S->rep[0]; D.rep1 = S->rep[1]; D.rep2 = S->rep[2];
    D.litPos = 0; D.done = 0;
    return 0;
}

// decode up to ZHIP_DEC_CHUNK sequences into recs (+ a sentinel record); whole wave 1 (ZSTD_decodeSequence :1228-1345).
// Per batch of 64 sequences:
//   pass 1, one lane: only what is inherently serial — look the three states up, add up the bits the sequence takes, fetch
//     the three next-state fields, note (position, states) in LDS: two dependent LDS round trips per sequence;
//   pass 2, one lane per sequence: value bits, literal / match lengths, raw offset codes; then the repeat-offset history is
//     resolved in order (a short wave-uniform loop), positions come from two wave prefix sums, every record is validated
//     (ZSTD_execSequence's checks :1025-1054) and stored.
// deferOff (the block-parallel decoder of zhip_decode_big.h): positions are block-relative and the offset history may be symbolic —
// offsets are validated later, when the block's place in the frame and its incoming history are known.
__device__ inline void dec_seq_chunk(DecShared* S, SeqDec& D, ZhipDSeq* recs, int buf, uint32_t nbSeq, uint32_t litSize,
                                     uint32_t dstCap, uint32_t dictLen, bool deferOff = false)
{
    uint32_t const lane = (uint32_t)lane_id();
    const lds_u32* const TL = (const lds_u32*)(uintptr_t)dec_tab(S, 0);     // entries as two dwords: [next | nbAdd << 16 | nb << 24], [base]
    const lds_u32* const TO = (const lds_u32*)(uintptr_t)dec_tab(S, 1);
    const lds_u32* const TM = (const lds_u32*)(uintptr_t)dec_tab(S, 2);
    const lds_u32* const R = (const lds_u32*)(uintptr_t)S->ring;
    uint32_t const want = nbSeq - D.done < ZHIP_DEC_CHUNK ? nbSeq - D.done : ZHIP_DEC_CHUNK;
    uint32_t n = 0, err = 0;
    uint32_t const endD = 8 * D.size;
    DPROF_LOCAL
    while (n < want && !err) {
        uint32_t const nb = want - n < 64 ? want - n : 64;
        // the batch reads at most 64 * 89 bits + one word beyond: keep that much staged, never overwrite what is still ahead
        if ((D.Dpos >> 5) + ZHIP_DEC_RING_WORDS / 2 >= D.wLoaded) {
            __builtin_amdgcn_wave_barrier();
            seq_ring_fill(S, D, D.wLoaded, ZHIP_DEC_RING_WORDS / 2);
            D.wLoaded += ZHIP_DEC_RING_WORDS / 2;
            __builtin_amdgcn_wave_barrier();
        }
        DPROF_L(0);
        // ---- pass 1: only what is inherently serial, ONE LDS round trip per sequence — the three table entries and a 128-bit
        // window of the stream at the sequence's start are requested together (both addresses are known from the previous
        // sequence); value bits (<= 63) + state bits (<= 26) + the start's bit offset (<= 31) always fit the window.  Every lane
        // runs the same chain on the same addresses (LDS broadcasts), so there is no divergence to manage and the loop counter
        // stays scalar; lane 0's copy is taken afterwards.
        uint32_t d = D.Dpos, sLL = D.sLL, sOF = D.sOF, sML = D.sML;
        {
            bool const chunkHasLast = D.done + n + nb == nbSeq;
            uint32_t const steps = chunkHasLast ? nb - 1 : nb;     // the block's last sequence updates no state (:1335)
            for (uint32_t j = 0; j < steps; j++) {
                uint32_t const w = (d >> 5) & (ZHIP_DEC_RING_WORDS - 1);
                uint32_t const eL = TL[2 * sLL], eO = TO[2 * sOF], eM = TM[2 * sML];
                uint32_t const r0 = R[w], r1 = R[w + 1], r2 = R[w + 2], r3 = R[w + 3];
                S->bat[j][0] = d; S->bat[j][1] = sLL; S->bat[j][2] = sOF; S->bat[j][3] = sML;
                uint32_t const sum = eL + eO + eM;                           // fields add without carrying into each other: next < 2^9, nbAdd sum <= 63, nb sum <= 26
                uint32_t const aSum = (sum >> 16) & 0xFF, nL = eL >> 24, nM = eM >> 24, nO = eO >> 24;
                uint32_t const k = (d & 31) + aSum;                          // state bits start k bits into the window, k <= 94
                uint32_t const hi = k < 32 ? r0 : (k < 64 ? r1 : r2), lo = k < 32 ? r1 : (k < 64 ? r2 : r3);
                uint32_t t = (uint32_t)(((((uint64_t)hi << 32) | lo) << (k & 31)) >> 32);
                sLL = (eL & 0xFFFF) + ((t >> 1) >> (31 - nL)); t <<= nL;
                sML = (eM & 0xFFFF) + ((t >> 1) >> (31 - nM)); t <<= nM;
                sOF = (eO & 0xFFFF) + ((t >> 1) >> (31 - nO));
                d += aSum + (sum >> 24);
            }
            if (chunkHasLast) {
                uint32_t const sum = TL[2 * sLL] + TO[2 * sOF] + TM[2 * sML];
                S->bat[steps][0] = d; S->bat[steps][1] = sLL; S->bat[steps][2] = sOF; S->bat[steps][3] = sML;
                d += (sum >> 16) & 0xFF;
            }
        }
        D.Dpos = __builtin_amdgcn_readfirstlane(d); D.sLL = __builtin_amdgcn_readfirstlane(sLL);
        D.sOF = __builtin_amdgcn_readfirstlane(sOF); D.sML = __builtin_amdgcn_readfirstlane(sML);
        __builtin_amdgcn_wave_barrier();
        DPROF_L(1);
        // ---- pass 2 (lane j = sequence j)
        bool const on = lane < nb;
        uint32_t ll = 0, ml = 0, offv = 0, sel = 4;            // sel: 4 = new offset in offv; 0..3 = repeat-offset selector (0: code 0)
        uint32_t ll0 = 0;
        if (on) {
            uint32_t const dj = S->bat[lane][0];
            uint32_t const a = S->bat[lane][1], b2 = S->bat[lane][2], c = S->bat[lane][3];
            uint32_t const eL = TL[2 * a], bL = TL[2 * a + 1], eO = TO[2 * b2], bO = TO[2 * b2 + 1], eM = TM[2 * c], bM = TM[2 * c + 1];
            uint32_t const aL = (eL >> 16) & 0xFF, aO = (eO >> 16) & 0xFF, aM = (eM >> 16) & 0xFF;
            uint32_t const xo = seq_field(R, dj, aO);
            ml = bM + seq_field(R, dj + aO, aM);
            ll = bL + seq_field(R, dj + aO + aM, aL);
            ll0 = bL == 0;
            if (aO > 1) { offv = bO + xo; sel = 4; }
            else if (aO == 0) sel = 0;
            else sel = bO + ll0 + xo;                          // 1..3
        }
        DPROF_L(2);
        // repeat-offset history in order (:1277-1300); all values wave-uniform
        uint32_t rep0 = D.rep0, rep1 = D.rep1, rep2 = D.rep2;
        {   // only sequences that USE the history are visited; the new offsets between two of them are pushed in one step
            unsigned long long reps = __ballot(on && sel != 4);
            uint32_t h = 0;                                     // history is current up to (not including) sequence h
            for (;;) {
                uint32_t const j = reps ? (uint32_t)first_lane(reps) : nb;
                uint32_t const m = j - h;                       // new offsets h .. j-1
                if (m) {
                    uint32_t const o1 = __builtin_amdgcn_readlane(offv, (int)j - 1);
                    uint32_t const o2 = m >= 2 ? __builtin_amdgcn_readlane(offv, (int)j - 2) : rep0;
                    uint32_t const o3 = m >= 3 ? __builtin_amdgcn_readlane(offv, (int)j - 3) : (m == 2 ? rep0 : rep1);
                    rep0 = o1; rep1 = o2; rep2 = o3;
                }
                if (!reps) break;
                reps &= reps - 1;
                uint32_t const sj = __builtin_amdgcn_readlane(sel, (int)j);
                uint32_t off;
                if (sj == 0) {
                    uint32_t const z = __builtin_amdgcn_readlane(ll0, (int)j);
                    off = z ? rep1 : rep0; rep1 = z ? rep0 : rep1; rep0 = off;
                } else {
                    uint32_t t = sj == 3 ? rep0 - 1 : (sj == 1 ? rep1 : rep2);
                    if (t == 0) t = 0xFFFFFFFFu;               // 0 is invalid: rejected by the offset check below
                    if (sj != 1) rep2 = rep1;
                    rep1 = rep0; rep0 = t; off = t;
                }
                if (lane == j) offv = off;
                h = j + 1;
            }
        }
        D.rep0 = rep0; D.rep1 = rep1; D.rep2 = rep2;
        DPROF_L(3);
        // positions: exclusive prefix sums of ll and ll + ml over the batch
        uint32_t incL = ll, incT = ll + ml;
        for (int sft = 1; sft < 64; sft <<= 1) {
            uint32_t const a = __shfl_up(incL, (unsigned)sft), b2 = __shfl_up(incT, (unsigned)sft);
            if ((int)lane >= sft) { incL += a; incT += b2; }
        }
        uint32_t const litPos = D.litPos + incL - ll;
        uint64_t const outPos64 = (uint64_t)D.outPos + (incT - (ll + ml));          // 64 * 2^18 fits, D.outPos may be near 2^32
        uint32_t e = 0;
        if (on) {
            if (ll > litSize || litPos > litSize - ll) e = ZHIP_DE_CORRUPT;
            else if (outPos64 + ll + ml > dstCap) e = ZHIP_DE_DST_SMALL;
            else if (!deferOff && (uint64_t)offv > outPos64 + ll + dictLen) e = ZHIP_DE_CORRUPT;
        }
        unsigned long long const bads = __ballot(e != 0);
        if (bads) { err = __builtin_amdgcn_readlane(e, first_lane(bads)); break; }
        if (on) { ZhipDSeq r; r.outPos = (uint32_t)outPos64; r.litPos = litPos; r.off = offv; r.ml = ml; recs[n + lane] = r; }
        D.litPos += __builtin_amdgcn_readlane(incL, (int)nb - 1);
        D.outPos += __builtin_amdgcn_readlane(incT, (int)nb - 1);
        n += nb;
        DPROF_L(4);
    }
    DPROF_LFLUSH(24);
    if (!err && D.done + n == nbSeq && D.Dpos != endD) err = ZHIP_DE_CORRUPT;          // BIT_endOfDStream (:1677)
    D.done += n;
    if (lane == 0) {
        ZhipDSeq r; r.outPos = D.outPos; r.litPos = D.litPos; r.off = 0; r.ml = 0; recs[err ? 0 : n] = r;
        S->cnt[buf] = err ? 0 : n; S->endOut = D.outPos; S->endLit = D.litPos;
        if (D.done == nbSeq && !err) { S->rep[0] = D.rep0; S->rep[1] = D.rep1; S->rep[2] = D.rep2; }
        if (err) atomicMax(&S->status, err);
    }
    __threadfence_block();
}

// ------------------------------------------------------------------ sequence execution, wave 0
struct LitSrc { const uint8_t* p; uint32_t mode; uint32_t byte; };      // mode 2: RLE (byte repeated)

// exact copy of n <= 64 bytes by ONE lane, source and destination disjoint: every load is issued before the first store
// (one memory round trip, not one per 8 bytes); the last chunk is re-based to end exactly at n (it rewrites equal bytes)
__device__ __forceinline__ void lane_copy(uint8_t* d, const uint8_t* s, uint32_t n)
{
    if (n >= 8) {
        uint64_t v[8]; uint32_t const lastOff = n - 8;
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) { uint32_t const o = 8 * k < lastOff ? 8 * k : lastOff; v[k] = 8 * k < n ? ld64(s + o) : 0; }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) { uint32_t const o = 8 * k < lastOff ? 8 * k : lastOff; if (8 * k < n) __builtin_memcpy(d + o, &v[k], 8); }
    } else if (n >= 4) {
        uint32_t const a = ld32(s), b2 = ld32(s + n - 4);
        __builtin_memcpy(d, &a, 4); __builtin_memcpy(d + n - 4, &b2, 4);
    } else if (n) {
        uint8_t const a = s[0], b2 = s[n >> 1], c = s[n - 1];
        d[0] = a; d[n >> 1] = b2; d[n - 1] = c;
    }
}
__device__ __forceinline__ void lane_fill(uint8_t* d, uint32_t byte, uint32_t n)
{
    uint64_t const v = 0x0101010101010101ull * byte; uint32_t i = 0;
    for (; i + 8 <= n; i += 8) __builtin_memcpy(d + i, &v, 8);
    for (; i < n; i++) d[i] = (uint8_t)byte;
}
// exact copy of n bytes by the whole wave (no overlap between source and destination)
__device__ __forceinline__ void wave_copy(uint8_t* d, const uint8_t* s, uint32_t n)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const full = n & ~7u;
    for (uint32_t i = 8 * lane; i < full; i += 512) { uint64_t v; __builtin_memcpy(&v, s + i, 8); __builtin_memcpy(d + i, &v, 8); }
    if (lane < (n & 7)) d[full + lane] = s[full + lane];
}
__device__ __forceinline__ void wave_fill(uint8_t* d, uint32_t byte, uint32_t n)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint64_t const v = 0x0101010101010101ull * byte; uint32_t const full = n & ~7u;
    for (uint32_t i = 8 * lane; i < full; i += 512) __builtin_memcpy(d + i, &v, 8);
    if (lane < (n & 7)) d[full + lane] = (uint8_t)byte;
}
// up to ZHIP_DEC_GROUP disjoint copies (each n >= 8 or n == 0 = unused slot) with every lane's loads of all of them in flight
// before the first store: one memory round trip for the group instead of one per copy.  The first 512 bytes of each; longer tails follow.
#define ZHIP_DEC_GROUP 4
__device__ __forceinline__ void wave_copy_x4(uint8_t* const d[ZHIP_DEC_GROUP], const uint8_t* const s[ZHIP_DEC_GROUP], const uint32_t n[ZHIP_DEC_GROUP])
{
    uint32_t const lane = (uint32_t)lane_id();
    uint64_t v[ZHIP_DEC_GROUP]; uint32_t o[ZHIP_DEC_GROUP]; bool has[ZHIP_DEC_GROUP];
#pragma unroll
    for (int g = 0; g < ZHIP_DEC_GROUP; g++) {
        has[g] = 8 * lane < n[g];
        o[g] = has[g] ? (8 * lane + 8 <= n[g] ? 8 * lane : n[g] - 8) : 0;     // the last chunk is re-based to end exactly at n
        v[g] = has[g] ? ld64(s[g] + o[g]) : 0;
    }
#pragma unroll
    for (int g = 0; g < ZHIP_DEC_GROUP; g++) if (has[g]) __builtin_memcpy(d[g] + o[g], &v[g], 8);
#pragma unroll
    for (int g = 0; g < ZHIP_DEC_GROUP; g++) if (n[g] > 512) wave_copy(d[g] + 512, s[g] + 512, n[g] - 512);
}
// periodic copy: d[k] = s[k % period], k < n (the overlapping-match case: the period bytes at s already exist)
__device__ __forceinline__ void wave_copy_periodic(uint8_t* d, const uint8_t* s, uint32_t period, uint32_t n)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t m = lane % period; uint32_t const stepm = 64 % period;
    for (uint32_t k = lane; k < n; k += 64) { d[k] = s[m]; m += stepm; if (m >= period) m -= period; }
}

// one hand-over chunk (ZSTD_execSequence :1001-1095), 64 sequences per step, one lane per sequence:
//   * every lane copies its own literals (their source and destination are known from the records);
//   * matches in rounds: a match is READY when its source lies entirely below the output of the first match that is still
//     pending (everything below that point is final) — ready matches are copied at once, each by its own lane (<= 64 bytes)
//     or by the whole wave (long, periodic = offset < length, or starting in the dictionary), then one s_waitcnt makes them
//     visible and the next round looks again.  The first pending match is always ready, so a round retires at least one;
//     far offsets retire a whole batch in one round.
// out = start of the frame's content; positions in the records are relative to it.
__device__ inline void dec_exec_chunk(const ZhipDSeq* recs, uint32_t cnt, uint8_t* out, LitSrc L, const uint8_t* dictEnd)
{
    uint32_t const lane = (uint32_t)lane_id();
    for (uint32_t b0 = 0; b0 < cnt; b0 += 64) {
        uint32_t const nb = cnt - b0 < 64 ? cnt - b0 : 64;
        bool const on = lane < nb;
        ZhipDSeq r; uint32_t nextLit;
        {   uint4 v = *(const uint4*)(recs + b0 + (on ? lane : 0)); r.outPos = v.x; r.litPos = v.y; r.off = v.z; r.ml = v.w;
            nextLit = recs[b0 + (on ? lane : 0) + 1].litPos; }
        uint32_t const ll = on ? nextLit - r.litPos : 0;
        // literals: short runs by their own lane, long runs by the whole wave
        if (ll && ll <= 64) { if (L.mode == 2) lane_fill(out + r.outPos, L.byte, ll); else lane_copy(out + r.outPos, L.p + r.litPos, ll); }
        unsigned long long longs = __ballot(ll > 64);
        while (longs) {
            if (L.mode == 2) {
                int const j = first_lane(longs); longs &= longs - 1;
                wave_fill(out + __builtin_amdgcn_readlane(r.outPos, j), L.byte, __builtin_amdgcn_readlane(ll, j));
                continue;
            }
            uint8_t* d4[ZHIP_DEC_GROUP]; const uint8_t* s4[ZHIP_DEC_GROUP]; uint32_t n4[ZHIP_DEC_GROUP];
#pragma unroll
            for (int g = 0; g < ZHIP_DEC_GROUP; g++) {
                if (longs) {
                    int const j = first_lane(longs); longs &= longs - 1;
                    d4[g] = out + __builtin_amdgcn_readlane(r.outPos, j); s4[g] = L.p + __builtin_amdgcn_readlane(r.litPos, j); n4[g] = __builtin_amdgcn_readlane(ll, j);
                } else { d4[g] = out; s4[g] = L.p; n4[g] = 0; }
            }
            wave_copy_x4(d4, s4, n4);
        }
        __threadfence_block();                                  // the batch's literals and everything before the batch are readable
        uint32_t const o = r.outPos + ll, off = r.off, ml = r.ml;           // my match: out[o .. o+ml) = virtual[o-off ..]
        bool const inDict = on && off > o;                      // starts before the frame (validated against the dictionary's length)
        uint32_t const back = inDict ? off - o : 0;
        // srcEnd: first position of out[] my match does NOT read (periodic matches read right up to their own start)
        uint32_t const srcEnd = !on ? 0 : inDict ? (ml > back ? ((ml < off ? ml : off) - back) : 0) : (off < ml ? o : o - off + ml);
        bool const simple = on && !inDict && off >= ml && ml <= 64;         // one lane, one round trip
        unsigned long long pending = __ballot(on);
        DPROF_ADD(30, 1);
        while (pending) {
            DPROF_ADD(29, 1);
            int const f = first_lane(pending);
            uint32_t const limit = __builtin_amdgcn_readlane(o, f);
            bool const ready = ((pending >> lane) & 1) && ((int)lane == f || srcEnd <= limit);
            if (ready && simple) lane_copy(out + o, out + o - off, ml);
            unsigned long long plain = __ballot(ready && !simple && !inDict && off >= ml);     // long, not periodic, inside the frame
            unsigned long long coop = __ballot(ready && !simple) & ~plain;
            while (plain) {
                uint8_t* d4[ZHIP_DEC_GROUP]; const uint8_t* s4[ZHIP_DEC_GROUP]; uint32_t n4[ZHIP_DEC_GROUP];
#pragma unroll
                for (int g = 0; g < ZHIP_DEC_GROUP; g++) {
                    if (plain) {
                        int const j = first_lane(plain); plain &= plain - 1;
                        uint32_t const oj = __builtin_amdgcn_readlane(o, j), offj = __builtin_amdgcn_readlane(off, j);
                        d4[g] = out + oj; s4[g] = out + oj - offj; n4[g] = __builtin_amdgcn_readlane(ml, j);
                    } else { d4[g] = out; s4[g] = out; n4[g] = 0; }
                }
                wave_copy_x4(d4, s4, n4);
            }
            while (coop) {
                int const j = first_lane(coop); coop &= coop - 1;
                uint32_t const oj = __builtin_amdgcn_readlane(o, j), offj = __builtin_amdgcn_readlane(off, j), mlj = __builtin_amdgcn_readlane(ml, j);
                uint8_t* const d = out + oj;
                if (offj > oj) {                                // starts in the dictionary (:1052-1066): byte k of the match is dict[dictLen - back + k]
                    // for k < back, then out[k - back] (k < off: bytes that exist), then periodic with period off
                    uint32_t const bk = offj - oj, nA = mlj < bk ? mlj : bk;
                    wave_copy(d, dictEnd - bk, nA);
                    if (mlj > bk) wave_copy(d + bk, out, (mlj < offj ? mlj : offj) - bk);
                    if (mlj > offj) { __threadfence_block(); wave_copy_periodic(d + offj, d, offj, mlj - offj); }
                } else if (offj >= mlj) wave_copy(d, d - offj, mlj);
                else wave_copy_periodic(d, d - offj, offj, mlj);
            }
            pending &= ~__ballot(ready);
            __threadfence_block();
        }
    }
}

// ------------------------------------------------------------------ one frame, whole workgroup
__device__ inline void decode_frame(DecShared* S, const uint8_t* src, uint32_t srcLen, uint8_t* out, uint32_t dstCap,
                                    uint8_t* litBuf, ZhipDSeq* recBuf, const ZhipDDictDev& dictArg /* the kernel argument itself: its fields stay scalar */, bool hasDict, const uint64_t* defTabs, ZhipDResult* res)
{
    // Control flow around the workgroup barriers is wave-uniform BY CONSTRUCTION (ZHIP_UNIFORM = v_readfirstlane): the wave index, every
    // status / count read back from LDS and hence every loop condition live in scalar registers and branch with s_cbranch_scc.  Left as
    // per-lane values (the compiler cannot prove an LDS load uniform) the loops became EXEC-masked, and in round 3 the compiler merged the
    // `tid == 0` epilogue of one frame with the `tid == 0` queue pop of the next into a lane-0-only path scheduled AFTER the other 127 lanes'
    // loop — they met the next barrier without it, read the stale queue slot and decoded frame 0 for ever.  ZHIP_CONVERGE()
    // (a convergent no-op) between two leader-only regions keeps the compiler from threading one into the other.
    uint32_t const tid = threadIdx.x, wave = ZHIP_UNIFORM(tid >> 6), lane = tid & 63;
    DPROF_BEGIN
    DecHeader const H = dec_frame_header(src, srcLen);
    const ZhipDDictDev* const dict = hasDict ? &dictArg : nullptr;
    bool const dictEnt = hasDict && dictArg.hasEntropy;        // wave-uniform (kernel argument)
    uint32_t const dictLen = dict ? dict->len : 0;
    const uint8_t* const dictEnd = dict ? dict->content + dict->len : nullptr;
    if (tid == 0) {
        uint32_t err = H.err;
        if (!err && H.dictID && (!dict || dict->dictID != H.dictID)) err = ZHIP_DE_DICT_WRONG;
        S->status = err; S->hufValid = 0; S->fseValid = 0; S->rep[0] = 1; S->rep[1] = 4; S->rep[2] = 8;
        if (dictEnt) { S->hufValid = 1; S->fseValid = 1; S->hufLog = dict->hufLog; for (int k = 0; k < 3; k++) { S->log[k] = dict->log[k]; S->rep[k] = dict->rep[k]; } }
    }
    ZHIP_CONVERGE();
    if (dictEnt) {                                              // the dictionary's tables, unless the previous frame left them untouched
        bool const needHuf = !ZHIP_UNIFORM(S->dictHufIn), needFse = !ZHIP_UNIFORM(S->dictFseIn);
        __syncthreads();
        if (needHuf) {
            if (dict->hufLog <= 11) { for (uint32_t i = tid; i < (1u << dict->hufLog); i += ZHIP_DEC_THREADS) ((uint32_t*)S->huf)[i] = dict->huf2[i]; }
            else for (uint32_t i = tid; i < (1u << dict->hufLog); i += ZHIP_DEC_THREADS) S->huf[i] = dict->huf[i];
        }
        if (needFse) for (uint32_t i = tid; i < 1280; i += ZHIP_DEC_THREADS) S->fseAll[i] = dict->fse[i];
        ZHIP_CONVERGE();
        if (tid == 0) { S->dictHufIn = 1; S->dictFseIn = 1; }
        ZHIP_CONVERGE();
    }
    __syncthreads();
    DPROF(wave ? 16 : 0);                                       // frame setup
    uint32_t status = ZHIP_UNIFORM(S->status);
    __syncthreads();                                            // everybody has read the status before anybody may raise it again (a wave that
                                                                // ran ahead and flagged an error must not split the workgroup's control flow)
    uint32_t ip = H.size, op = 0;
    SeqDec D; D.done = 0; D.outPos = 0; D.litPos = 0; D.sLL = D.sOF = D.sML = 0; D.rep0 = D.rep1 = D.rep2 = 0; D.base = src; D.size = 0; D.Dpos = 0; D.wLoaded = 0;
    bool last = false;
    while (!status && !last) {
        if (srcLen - ip < 3) { status = ZHIP_DE_SRC_WRONG; break; }
        uint32_t const bh = ZHIP_UNIFORM(src[ip] | (src[ip + 1] << 8) | (src[ip + 2] << 16));
        uint32_t const type = (bh >> 1) & 3, bsize = bh >> 3;
        last = bh & 1;
        ip += 3;
        if (type == 3) { status = ZHIP_DE_CORRUPT; break; }
        uint32_t const csize = type == 1 ? 1 : bsize;
        if (csize > srcLen - ip) { status = ZHIP_DE_SRC_WRONG; break; }
        if (type != 2) {                                       // raw / RLE block: the whole workgroup
            if (bsize > dstCap - op) { status = ZHIP_DE_DST_SMALL; break; }
            if (type == 0) {
                uint32_t const full = bsize & ~7u;
                for (uint32_t i = 8 * tid; i < full; i += 8 * ZHIP_DEC_THREADS) { uint64_t v; __builtin_memcpy(&v, src + ip + i, 8); __builtin_memcpy(out + op + i, &v, 8); }
                if (tid < (bsize & 7)) out[op + full + tid] = src[ip + full + tid];
            } else {
                uint64_t const v = 0x0101010101010101ull * src[ip]; uint32_t const full = bsize & ~7u;
                for (uint32_t i = 8 * tid; i < full; i += 8 * ZHIP_DEC_THREADS) __builtin_memcpy(out + op + i, &v, 8);
                if (tid < (bsize & 7)) out[op + full + tid] = src[ip];
            }
            __threadfence_block();
            __syncthreads();
            op += bsize; ip += csize;
            continue;
        }
        // compressed block (ZSTD_decompressBlock_internal, zstd_decompress_block.c:2072-2180)
        if (csize > H.blockMax) { status = ZHIP_DE_SRC_WRONG; break; }
        const uint8_t* const blk = src + ip;
        if (wave == 0) { dec_literals(S, blk, csize, H.blockMax, litBuf, dstCap - op); DPROF(1); }
        else {
            LitHeader const lh = dec_lit_header(blk, csize, H.blockMax);
            uint32_t err = lh.err ? 1u : 0u;                    // wave 0 reports the literal header's own errors
            if (!err) {
                uint32_t const secOff = lh.lh + lh.cSize;
                D.outPos = op; D.done = 0;
                uint32_t nbSeqW = 0;
                uint32_t const e2 = dec_seq_setup(S, blk + secOff, csize - secOff, D, defTabs, &nbSeqW);
                DPROF(17);
                if (e2) { if (lane == 0) atomicMax(&S->status, e2); }
                else if (nbSeqW) dec_seq_chunk(S, D, recBuf, 0, nbSeqW, lh.litSize, dstCap, dictLen);
                DPROF(18);
            } else if (lane == 0) S->nbSeq = 0;
        }
        __syncthreads();
        DPROF(wave ? 19 : 2);                                   // waiting for the other wave
        status = ZHIP_UNIFORM(S->status);
        __syncthreads();                                        // (same rule: read, then barrier, then the next writer)
        if (status) break;
        uint32_t const nbSeq = ZHIP_UNIFORM(S->nbSeq), litSize = ZHIP_UNIFORM(S->litSize);
        LitSrc L; L.mode = ZHIP_UNIFORM(S->litMode); L.byte = ZHIP_UNIFORM(S->litByte); L.p = L.mode == 1 ? blk + ZHIP_UNIFORM(S->litSrcOff) : litBuf;
        uint32_t const nChunks = (nbSeq + ZHIP_DEC_CHUNK - 1) / ZHIP_DEC_CHUNK;
        for (uint32_t c = 0; c < nChunks && !status; c++) {
            if (wave == 0) dec_exec_chunk(recBuf + (size_t)(c & 1) * (ZHIP_DEC_CHUNK + 1), ZHIP_UNIFORM(S->cnt[c & 1]), out, L, dictEnd);
            else if (c + 1 < nChunks) dec_seq_chunk(S, D, recBuf + (size_t)((c + 1) & 1) * (ZHIP_DEC_CHUNK + 1), (int)((c + 1) & 1), nbSeq, litSize, dstCap, dictLen);
            DPROF(wave ? 20 : 3);                               // executing / decoding a chunk
            __syncthreads();
            DPROF(wave ? 21 : 4);
            status = ZHIP_UNIFORM(S->status);
            __syncthreads();
        }
        if (status) break;
        {   // last literals (zstd_decompress_block.c:1681-1690)
            uint32_t const endOut = nbSeq ? ZHIP_UNIFORM(S->endOut) : op, endLit = nbSeq ? ZHIP_UNIFORM(S->endLit) : 0;
            uint32_t const rest = litSize - endLit;
            if (rest > dstCap - endOut) { status = ZHIP_DE_DST_SMALL; break; }
            if (wave == 0) { if (L.mode == 2) wave_fill(out + endOut, L.byte, rest); else wave_copy(out + endOut, L.p + endLit, rest); __threadfence_block(); }
            op = endOut + rest;
        }
        ip += csize;
        DPROF_ADD(8, nbSeq); DPROF_ADD(9, litSize);
        __syncthreads();
        DPROF(wave ? 22 : 5);                                   // last literals
    }
    if (!status && H.fcs != ~0ull && H.fcs != (uint64_t)op) status = ZHIP_DE_CORRUPT;
    uint32_t ck = 0;
    if (!status && H.checksum) { if (srcLen - ip < 4) status = ZHIP_DE_CHECKSUM; else ck = ld32(src + ip); }
    __syncthreads();
    if (tid == 0) { res->status = status; res->size = status ? 0 : op; res->hasChecksum = !status && H.checksum; res->checksum = ck; }
    ZHIP_CONVERGE();                                            // the caller's next leader-only region (the queue pop) stays a region of its own
    DPROF(wave ? 23 : 6);
}
#endif  // ZHIP_DECODE_HOST_ONLY

}  // namespace zhip
