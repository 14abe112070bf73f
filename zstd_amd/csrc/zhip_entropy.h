// zhip_entropy.h — gfx950 entropy stage + frame assembly: one 256-thread workgroup per unit.
//
// WHAT: from the match finder's sequences it produces the complete frame the reference's ZSTD_compress2 emits for
// that unit: frame header (lib/compress/zstd_compress.c:4626-4672), block header (:4586-4590), literals section
// (zstd_compress_literals.c:129-235 -> huf_compress.c), sequences section (zstd_compress.c:2934-2997 ->
// zstd_compress_sequences.c, fse_compress.c), with every fallback (raw / RLE literals, predefined / RLE / FSE
// tables, raw block) decided exactly as the reference decides it.
//
// HOW (CDNA4): the data-parallel parts run on all 4 wavefronts — literal gather + byte histogram with LDS atomics,
// Huffman bit packing (one wavefront per huff0 stream: per-lane runs, wave prefix-scan of bit lengths, direct packing
// into the output with atomicOr only on run-boundary words), sequence code histograms, sequence bit packing
// (block prefix-scan of per-sequence bit counts).  The inherently ordered parts (code-length assignment, FSE
// normalisation/table build, the three FSE state chains) run on single lanes of different wavefronts concurrently.
#pragma once
#include <hip/hip_runtime.h>
#include "zhip_common.h"
#include "zhip_tables.h"

namespace zhip {

#define ZHIP_ENT_THREADS 256

// ------------------------------------------------------------------ constant tables (zstd_internal.h:123-168)
__device__ static const uint8_t kLLbits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3, 4,6,7,8,9,10,11,12, 13,14,15,16 };
__device__ static const uint8_t kMLbits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                                1,1,1,1,2,2,3,3, 4,4,5,7,8,9,10,11, 12,13,14,15,16 };
__device__ static const int16_t kLLnorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1, 2,2,2,2,2,2,2,2, 2,3,2,1,1,1,1,1, -1,-1,-1,-1 };
__device__ static const int16_t kMLnorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                                1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
__device__ static const int16_t kOFnorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };
__device__ static const uint8_t kLLcode[64] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
    22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 };
__device__ static const uint8_t kMLcode[128] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
    32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
    40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
    42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42 };

// the four small tables are copied to LDS once per workgroup (a lookup in the global copy costs an HBM/L2 round trip)
// -log2(x / 256) in 1/256 bit units, x = 1..255 (lib/compress/zstd_compress_sequences.c:21-44 kInverseProbabilityLog256)
__device__ static const uint16_t kInvProbLog256[256] = {
    0,    2048, 1792, 1642, 1536, 1453, 1386, 1329, 1280, 1236, 1197, 1162, 1130, 1100, 1073, 1047,
    1024, 1001, 980,  960,  941,  923,  906,  889,  874,  859,  844,  830,  817,  804,  791,  779,
    768,  756,  745,  734,  724,  714,  704,  694,  685,  676,  667,  658,  650,  642,  633,  626,
    618,  610,  603,  595,  588,  581,  574,  567,  561,  554,  548,  542,  535,  529,  523,  517,
    512,  506,  500,  495,  489,  484,  478,  473,  468,  463,  458,  453,  448,  443,  438,  434,
    429,  424,  420,  415,  411,  407,  402,  398,  394,  390,  386,  382,  377,  373,  370,  366,
    362,  358,  354,  350,  347,  343,  339,  336,  332,  329,  325,  322,  318,  315,  311,  308,
    305,  302,  298,  295,  292,  289,  286,  282,  279,  276,  273,  270,  267,  264,  261,  258,
    256,  253,  250,  247,  244,  241,  239,  236,  233,  230,  228,  225,  222,  220,  217,  215,
    212,  209,  207,  204,  202,  199,  197,  194,  192,  190,  187,  185,  182,  180,  178,  175,
    173,  171,  168,  166,  164,  162,  159,  157,  155,  153,  151,  149,  146,  144,  142,  140,
    138,  136,  134,  132,  130,  128,  126,  123,  121,  119,  117,  115,  114,  112,  110,  108,
    106,  104,  102,  100,  98,   96,   94,   93,   91,   89,   87,   85,   83,   82,   80,   78,
    76,   74,   73,   71,   69,   67,   66,   64,   62,   61,   59,   57,   55,   54,   52,   50,
    49,   47,   46,   44,   42,   41,   39,   37,   36,   34,   33,   31,   30,   28,   26,   25,
    23,   22,   20,   19,   17,   16,   14,   13,   11,   10,   8,    7,    5,    4,    2,    1,
};
struct CodeTabs { uint8_t llCode[64]; uint8_t mlCode[128]; uint8_t llBits[36]; uint8_t mlBits[56]; };
__device__ __forceinline__ uint32_t ll_code(const CodeTabs& T, uint32_t ll) { return ll > 63 ? hb32(ll) + 19 : T.llCode[ll]; }          // internal.h:520
__device__ __forceinline__ uint32_t ml_code(const CodeTabs& T, uint32_t mlBase) { return mlBase > 127 ? hb32(mlBase) + 36 : T.mlCode[mlBase]; }   // :537

// ------------------------------------------------------------------ LDS layout of the workgroup
struct EntShared {
    // The Huffman builder's workspace (8.6 KB, wave 0, phase B only) lies over what is dead or not yet alive while it runs: the sampling
    // histograms (read at the start of the literals job, before the builder) and the scan area (phase C).  The four per-wavefront literal
    // histograms stay alive through phase B (round 6): wavefront w counts the bytes of Huffman stream w, so the streams' sizes are
    // sum(count x code length) — no sizing pass over the literals.  21.5 KB: seven workgroups per CU (its registers allow six).
    uint32_t hist[4][256];        // per-wavefront = per-stream literal histograms; hist[0] becomes the block's (sum of the four)
    union {
        struct {
            uint32_t sampleHist[2][256];      // directly behind hist[]: the two together are the packers' LDS images in phase C
            uint32_t scan[ZHIP_ENT_THREADS + 8];
        };
        HufWork  huf;
    };
    uint32_t code[256];           // huff0 code: value << 8 | nbBits
    uint32_t seqCount[3][64];     // LL / OF / ML code histograms
    int16_t  norm[3][56];
    FseCTable ct[3];              // LL, OF, ML
    alignas(16) uint8_t symScratch[3][512];       // fse_build_ctable_wave's scratch
    uint16_t cumul[3][64];
    uint8_t  ncount[3][64];       // NCount header bytes (or the RLE byte)
    uint32_t ncWords[3][16];      // their bit-level assembly area (fse_write_ncount_wave)
    uint32_t ncountSize[3];
    uint32_t encType[3];          // set_basic 0 / set_rle 1 / set_compressed 2 / set_repeat 3 (zstd_internal.h:102)
    uint32_t maxCode[3];
    uint32_t finalState[3];
    uint32_t chainBits[3], extraBits[3];   // per table: the state bits of all chain records / the extra bits of all sequences (the bitstream's size without a pass over it)
    uint8_t  hufHdr[136];
    // scalars broadcast through LDS
    uint32_t litSize, hufHdrSize, huffLog, litMode /*0 raw,1 rle,2 huf*/, singleStream, litType /*2 compressed, 3 repeat*/, hufMaxSym /* of a new table */;
    uint32_t streamBits[4], streamBytes[4], streamOff[4];
    uint32_t litSectionSize, seqSectionSize, seqBitsTotal, failRaw;
    CodeTabs tabs;
};

// The same workspace for the one-wavefront form (NT = 64): one histogram instead of four, no cross-wave scan, no sampling
// histograms (the sampling heuristic needs 40 KB of literals; units of this form are at most ZHIP_ENT_SMALL_MAX bytes), and the
// Huffman builder's workspace shares its bytes with the sequence tables — the single wavefront is done with the literals job before
// it starts the first table.  12.6 KB: twelve records resident per CU.
#define ZHIP_ENT_SMALL_MAX 8192u
struct EntSharedSmall {
    uint32_t hist[1][256];
    uint32_t code[256];
    uint32_t scan[8];
    uint32_t seqCount[3][64];
    union {
        HufWork huf;
        struct { FseCTable ct[3]; int16_t norm[3][56]; alignas(16) uint8_t symScratch[1][512]; uint16_t cumul[3][64]; };       // ONE scratch: the single wavefront builds its tables one after the other
    };
    uint8_t  ncount[3][64];
    uint32_t ncWords[3][16];
    uint32_t ncountSize[3];
    uint32_t encType[3];
    uint32_t maxCode[3];
    uint32_t finalState[3];
    uint32_t chainBits[3], extraBits[3];   // per table: the state bits of all chain records / the extra bits of all sequences (the bitstream's size without a pass over it)
    uint8_t  hufHdr[136];
    uint32_t litSize, hufHdrSize, huffLog, litMode, singleStream, litType, hufMaxSym;
    uint32_t streamBits[4], streamBytes[4], streamOff[4];
    uint32_t litSectionSize, seqSectionSize, seqBitsTotal, failRaw;
    uint32_t sampleHist[2][1];    // never touched (see above); keeps the shared code one text
    CodeTabs tabs;
};

// ------------------------------------------------------------------ small block-wide helpers
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    int const lane = (int)(threadIdx.x & 63);
    for (int d = 1; d < 64; d <<= 1) { uint32_t const o = __shfl_up(v, (unsigned)d); if (lane >= d) v += o; }
    return v;
}
// exclusive prefix sum over the NT threads of the workgroup (thread order); *total gets the grand total. uses sh->scan
template <uint32_t NT, typename SH>
__device__ inline uint32_t block_excl_scan(SH* sh, uint32_t v, uint32_t* total)
{
    int const t = (int)threadIdx.x, lane = t & 63, w = t >> 6;
    uint32_t const inc = wave_incl_scan(v);
    if (NT == 64) { *total = __shfl(inc, 63); return inc - v; }
    if (lane == 63) sh->scan[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int i = 0; i < (int)(NT / 64); i++) { if (i < w) base += sh->scan[i]; tot += sh->scan[i]; }
    *total = tot;
    __syncthreads();
    return base + inc - v;
}

// zero exactly [p, p+nbytes) with all threads of the workgroup (neighbouring bytes belong to finished sections)
template <uint32_t NT>
__device__ inline void zero_bytes(uint8_t* p, uint32_t nbytes)
{
    uint32_t const t = threadIdx.x;
    uint32_t head = (4u - (uint32_t)((uintptr_t)p & 3)) & 3u; if (head > nbytes) head = nbytes;
    if (t < head) p[t] = 0;                                              // head, tail < 4 <= NT
    uint32_t* const w = (uint32_t*)(p + head);
    uint32_t const words = (nbytes - head) >> 2;
    for (uint32_t i = t; i < words; i += NT) w[i] = 0;
    uint32_t const tail = nbytes - head - 4 * words;
    if (t < tail) p[head + 4 * words + t] = 0;
}

// OR `nbits` (<= 64 - (bitpos & 31)) bits of `v` into the LSB-first bit string that starts at word pointer `w32`
// (bit 0 of the string = bit 0 of w32[0]).  Output words were zeroed beforehand; neighbouring lanes share boundary
// words, hence atomics.  Used for run-boundary words only; interior words are written by plain stores.
__device__ __forceinline__ void or_bits(uint32_t* w32, uint64_t bitpos, uint64_t v, uint32_t nbits)
{
    if (!nbits) return;
    uint32_t* p = w32 + (bitpos >> 5);
    uint32_t const sh = (uint32_t)(bitpos & 31);
    atomicOr(p, (uint32_t)(v << sh));
    if (sh + nbits > 32) {
        uint64_t const rest = v >> (32 - sh);
        atomicOr(p + 1, (uint32_t)rest);
        if (sh + nbits > 64) atomicOr(p + 2, (uint32_t)(rest >> 32));
    }
}

// A per-lane LSB-first packer that owns the bit range [start, start+len) of a shared bit string in global memory
// (words zeroed beforehand).  The accumulator starts with (start & 31) zero bits, so every flush is a whole 32-bit word
// at a word-aligned position: the first and the last (partial) words are shared with the neighbouring lanes' ranges
// and go through atomicOr, the words in between are plain stores.  add() takes at most 31 bits.
struct RunPacker {
    uint32_t* wp; uint64_t acc; uint32_t nb; bool first;
    __device__ __forceinline__ void init(uint32_t* base, uint64_t startBit) { wp = base + (startBit >> 5); nb = (uint32_t)startBit & 31; acc = 0; first = true; }
    __device__ __forceinline__ void add(uint32_t v, uint32_t n)
    {
        acc |= (uint64_t)v << nb; nb += n;
        if (nb >= 32) {
            if (first) atomicOr(wp, (uint32_t)acc); else *wp = (uint32_t)acc;
            first = false; wp++; acc >>= 32; nb -= 32;
        }
    }
    __device__ __forceinline__ void finish() { if (nb) atomicOr(wp, (uint32_t)acc); nb = 0; acc = 0; }
};

// The same with the interior words collected four at a time and stored as 16 bytes (round 5).  A lane of the full-size huff0 packer owns a
// run of ~190 symbols = ~48 output words of its own: stored one by one, every store instruction of the wavefront touched 64 different
// lines four bytes at a time, the lines left L2 between two visits, and WRITE_SIZE of the stage was 7 x the frames it wrote
// (profiles/r04_L1_datagen_sq_tcc.txt).
struct RunPacker4 {
    uint32_t* wp; uint64_t acc; uint32_t nb; bool first;
    uint32_t b0, b1, b2, b3, nbuf;
    __device__ __forceinline__ void init(uint32_t* base, uint64_t startBit) { wp = base + (startBit >> 5); nb = (uint32_t)startBit & 31; acc = 0; first = true; nbuf = 0; b0 = b1 = b2 = b3 = 0; }
    __device__ __forceinline__ void add(uint32_t v, uint32_t n)
    {
        acc |= (uint64_t)v << nb; nb += n;
        if (nb >= 32) {
            uint32_t const w = (uint32_t)acc;
            acc >>= 32; nb -= 32;
            if (first) { atomicOr(wp, w); first = false; wp++; return; }
            if (nbuf == 0) b0 = w; else if (nbuf == 1) b1 = w; else if (nbuf == 2) b2 = w; else b3 = w;
            if (++nbuf == 4) { uint4 const q = { b0, b1, b2, b3 }; __builtin_memcpy(wp, &q, 16); wp += 4; nbuf = 0; }
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nbuf >= 1) wp[0] = b0;
        if (nbuf >= 2) wp[1] = b1;
        if (nbuf >= 3) wp[2] = b2;
        wp += nbuf; nbuf = 0;
        if (nb) atomicOr(wp, (uint32_t)acc);
        nb = 0; acc = 0;
    }
};

// ------------------------------------------------------------------ sequence field access
__device__ __forceinline__ void seq_fields(const ZhipSeq* seqs, const ZhipParse& m, uint32_t i, uint32_t& ll, uint32_t& mlBase, uint32_t& offBase)
{
    ZhipSeq const s = seqs[i];
    ll = s.litLength; mlBase = s.mlBase; offBase = s.offBase;
    if (i == m.longPos) { if (m.longType == 1) ll += 0x10000; else if (m.longType == 2) mlBase += 0x10000; }
}

__device__ __forceinline__ void seq_unpack(const ZhipSeq& s, const ZhipParse& m, uint32_t i, uint32_t& ll, uint32_t& mlBase, uint32_t& offBase)
{
    ll = s.litLength; mlBase = s.mlBase; offBase = s.offBase;
    if (i == m.longPos) { if (m.longType == 1) ll += 0x10000; else if (m.longType == 2) mlBase += 0x10000; }
}

// ------------------------------------------------------------------ frame / block headers
__device__ inline uint32_t dict_id_bytes(uint32_t dictID) { uint32_t const c = (dictID > 0) + (dictID >= 256) + (dictID >= 65536); return c == 3 ? 4 : c; }
__device__ inline uint32_t frame_header_size(uint32_t n, uint32_t dictID) { return 4 + 1 + dict_id_bytes(dictID) + (n < 256 ? 1 : (n < 65536 + 256 ? 2 : 4)); }
// zstd_compress.c:4626-4672 for a single-segment frame with content size, no checksum (library defaults), optional dictID
__device__ inline uint32_t write_frame_header(uint8_t* op, uint32_t n, uint32_t dictID, bool checksum = false)
{
    uint32_t const fcs = (n >= 256) + (n >= 65536 + 256);
    uint32_t const dcode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
    uint32_t pos = 5;
    op[0] = 0x28; op[1] = 0xB5; op[2] = 0x2F; op[3] = 0xFD;
    op[4] = (uint8_t)(dcode + ((checksum ? 1u : 0u) << 2) + (1u << 5) + (fcs << 6));       // :4637
    for (uint32_t i = 0; i < dict_id_bytes(dictID); i++) op[pos++] = (uint8_t)(dictID >> (8 * i));     // :4651-4657
    if (fcs == 0) { op[pos] = (uint8_t)n; return pos + 1; }
    if (fcs == 1) { uint32_t const v = n - 256; op[pos] = (uint8_t)v; op[pos + 1] = (uint8_t)(v >> 8); return pos + 2; }
    op[pos] = (uint8_t)n; op[pos + 1] = (uint8_t)(n >> 8); op[pos + 2] = (uint8_t)(n >> 16); op[pos + 3] = (uint8_t)(n >> 24); return pos + 4;
}

// ------------------------------------------------------------------ wide memory helpers
__device__ __forceinline__ uint64_t eld64(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__device__ __forceinline__ void est64(uint8_t* p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
// the (at most 8) bytes of src[pos .. n) as a little-endian word, never reading outside src[0 .. n)
__device__ __forceinline__ uint64_t eld64_in(const uint8_t* src, uint32_t pos, uint32_t n)
{
    if (pos + 8 <= n) return eld64(src + pos);
    if (n >= 8) { uint32_t const sh = pos + 8 - n; return sh >= 8 ? 0 : eld64(src + (n - 8)) >> (8 * sh); }
    uint64_t v = 0;
    for (uint32_t i = pos; i < n; i++) v |= (uint64_t)src[i] << (8 * (i - pos));
    return v;
}
__device__ __forceinline__ void hist8(uint32_t* h, uint64_t v, uint32_t cnt)
{
    for (uint32_t b = 0; b < cnt; b++) atomicAdd(&h[(uint32_t)(v >> (8 * b)) & 0xFF], 1u);
}

// one step of an FSE state chain (FSE_encodeSymbol, lib/common/fse.h:463): returns the record
// (nbBits << 12 | low bits of the state) and advances the state.  dBits/dFind are the symbol's transform, fetched by
// the caller ahead of time so that the state-table read is the only LDS access on the dependent path.
__device__ __forceinline__ uint32_t fse_chain_step(const FseCTable* ct, uint32_t& state, uint32_t dBits, int32_t dFind)
{
    uint32_t const nbOut = (state + dBits) >> 16;
    uint32_t const rec = (nbOut << 12) | (state & ((1u << nbOut) - 1));
    state = ct->state[(state >> nbOut) + dFind];
    return rec;
}

// ------------------------------------------------------------------ the FSE state chain of one table, by the whole wavefront
// The chain is a recurrence over the sequences, last to first: record_i = the low nbBits of the state, state = next(state, code_i).
// It is serial in principle, but an FSE state forgets its past — every step shifts nbBits of it out — so the chain is cut into 64
// slices: lane L runs its slice after a short warm-up on the entries above it from an arbitrary state, and is then CHECKED against
// the true state the lane above hands down.  Forgetting can be slow (a table with one dominant symbol emits ~0 bits per step: half
// of a text unit's slices are still off after 24 entries), so a lane whose entering state was wrong redoes its slice from the right
// one — but only until its new trajectory meets the old one (the state after every 8-entry chunk of the first run is kept in
// `ckpt`), after which everything, the state handed down included, is as before.  Rounds repeat until no lane changes: none or a few
// short ones in practice, as many full ones as lanes when nothing ever merges (= the serial chain).  Exact by construction.
// Then every slice is run once more from its true entering state, this time writing the records over the codes.
// Entries move 8 at a time (16-byte loads / stores per lane).
#define ZHIP_FSE_WARM 3u          /* warm-up, in 8-entry chunks */
__device__ __forceinline__ void fse_chain_chunk(const FseCTable* ct, uint16_t* arr, uint32_t c, uint32_t M, uint4 w, uint32_t& state, bool record, uint32_t& bits /* += nbBits of the records written */)
{   // entries [8c, 8c+8) ∩ [0, M), highest first; w = the 16 bytes at arr + 8c
    uint32_t x[4] = { w.x, w.y, w.z, w.w };                                   // two 16-bit entries per word; all indices below are static
    uint32_t const top = M - 8u * c < 8 ? M - 8u * c : 8;
    if (top == 8) {
        // a full chunk: the eight symbols' table parameters do not depend on the state, so they are fetched together and the
        // serial part is ONE dependent LDS read (the next state) per step
        uint32_t db[8]; int32_t df[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { uint32_t const sy = (x[q >> 1] >> ((q & 1) ? 16 : 0)) & 0xFFFFu; db[q] = ct->dBits[sy]; df[q] = ct->dFind[sy]; }
#pragma unroll
        for (int q = 7; q >= 0; q--) {
            uint32_t const sh16 = (q & 1) ? 16u : 0u;
            uint32_t const rec = fse_chain_step(ct, state, db[q], df[q]);
            x[q >> 1] = (x[q >> 1] & ~(0xFFFFu << sh16)) | (rec << sh16);
            if (record) bits += rec >> 12;
        }
        if (record) { w.x = x[0]; w.y = x[1]; w.z = x[2]; w.w = x[3]; __builtin_memcpy(arr + 8u * c, &w, 16); }
        return;
    }
#pragma unroll
    for (int q = 7; q >= 0; q--) if ((uint32_t)q < top) {
        uint32_t const sh16 = (q & 1) ? 16u : 0u;
        uint32_t const sy = (x[q >> 1] >> sh16) & 0xFFFFu;
        uint32_t const rec = fse_chain_step(ct, state, ct->dBits[sy], ct->dFind[sy]);
        x[q >> 1] = (x[q >> 1] & ~(0xFFFFu << sh16)) | (rec << sh16);
        if (record) bits += rec >> 12;
    }
    if (record) {
#pragma unroll
        for (int q = 0; q < 8; q++) if ((uint32_t)q < top) arr[8u * c + q] = (uint16_t)(x[q >> 1] >> ((q & 1) ? 16 : 0));
    }
}
// chunks hiC-1 down to loC through `state`, the next chunk's 16 bytes in flight while this one is walked.
// mode 0: nothing kept; 1: the state after each chunk goes to ckpt; 2: it is compared with ckpt — equal = the trajectory met the
// earlier one, stop and return true — and replaces it; 3: records are written over the codes
enum { ZC_DRY = 0, ZC_KEEP = 1, ZC_MEET = 2, ZC_RECORD = 3 };
__device__ __forceinline__ bool fse_chain_run(const FseCTable* ct, uint16_t* arr, uint32_t M, uint32_t hiC, uint32_t loC, uint32_t& state,
                                              int mode, uint16_t* ckpt, uint32_t lane, uint32_t& bits)
{
    if (hiC <= loC) return false;
    uint4 nxt; __builtin_memcpy(&nxt, arr + 8u * (hiC - 1), 16);
    uint32_t t = 0;
    for (uint32_t c = hiC; c > loC; c--, t++) {
        uint4 const cur = nxt;
        if (c - 1 > loC) __builtin_memcpy(&nxt, arr + 8u * (c - 2), 16);
        fse_chain_chunk(ct, arr, c - 1, M, cur, state, mode == ZC_RECORD, bits);
        if (mode == ZC_KEEP) ckpt[t * 64 + lane] = (uint16_t)state;
        else if (mode == ZC_MEET) {
            if (ckpt[t * 64 + lane] == (uint16_t)state) return true;
            ckpt[t * 64 + lane] = (uint16_t)state;
        }
    }
    return false;
}
// room `ckpt` needs, in uint16 entries, for a chain of M entries cut over G lanes
__host__ __device__ inline uint32_t fse_chain_ckpt_entries(uint32_t M, uint32_t G = 64) { uint32_t const chunks = (M + 7) >> 3; return 64u * ((chunks + G - 1) / G); }
// arr[0 .. M) codes -> records; returns (in every lane of the group) the final state (after entry 0).  lastCode = the code of
// sequence M (it only seeds the state).  The wavefront is cut into groups of G lanes (G = 64: one chain by the whole wavefront;
// G = 21: three chains — LL, OF, ML of one small block — side by side, lane 63 idle); ct / arr / lastCode are per GROUP (every
// lane passes its group's), M is the same for all.  ckpt: fse_chain_ckpt_entries(M, G) uint16 of scratch in global memory, private
// to this wavefront (all groups index it by t * 64 + lane); touched only when a lane owns two or more chunks (M > 8 G).
template <uint32_t G>
__device__ inline uint32_t fse_chain_wave(const FseCTable* ct, uint16_t* arr, uint32_t M, uint32_t lastCode, uint16_t* ckpt, uint32_t* recBits = nullptr /* the records' nbBits, summed over the wavefront (G = 21: the three chains together) */)
{
    uint32_t bits = 0;
    uint32_t const lane = (uint32_t)(threadIdx.x & 63);
    uint32_t const grp = lane / G, gl = lane - grp * G;                     // group, lane inside the group
    bool const inGroup = G == 64 || grp < 64 / G;                           // G = 21: lane 63 belongs to no chain
    uint32_t const init = fse_init_state2(ct, lastCode);
    if (M == 0) { if (recBits) *recBits = 0; return init; }
    uint32_t const chunks = (M + 7) >> 3, cpl = (chunks + G - 1) / G;       // chunks per lane
    uint32_t const used = (chunks + cpl - 1) / cpl;                         // lanes of a group that own something; lane 0 of a group owns the TOP slice
    bool const mine = inGroup && gl < used;
    uint32_t const topC = mine ? chunks - gl * cpl : 0;                     // own chunks [botC, topC)
    uint32_t const botC = topC > cpl ? topC - cpl : 0;
    // pass 1, nothing written to arr: warm-up (one that reaches the very top starts from the true state), then the slice
    uint32_t enter = init;
    if (mine && gl) {
        uint32_t w1 = topC + ZHIP_FSE_WARM;
        if (w1 >= chunks) w1 = chunks; else enter = 1u << ct->tableLog;
        fse_chain_run(ct, arr, M, w1, topC, enter, ZC_DRY, ckpt, lane, bits);
    }
    uint32_t fin = enter;
    bool const keep = cpl >= 2;                                               // one chunk per lane: nothing to meet (and a small record's output room is not borrowed)
    fse_chain_run(ct, arr, M, topC, botC, fin, keep ? ZC_KEEP : ZC_DRY, ckpt, lane, bits);
    // hand-down rounds
    for (;;) {
        uint32_t const above = __shfl_up(fin, 1);
        bool const bad = mine && gl && enter != above;
#ifdef ZHIP_CHAIN_DEBUG
        { unsigned long long const bb = __ballot(bad); if (lane == 0 && bb) printf("CHAIN M=%u tableLog=%u cpl=%u used=%u bad=%d lanes\n", M, ct->tableLog, cpl, used, (int)__builtin_popcountll(bb)); }
#endif
        if (!__ballot(bad)) break;
        if (bad) {
            enter = above;
            uint32_t st = above;
            if (!fse_chain_run(ct, arr, M, topC, botC, st, keep ? ZC_MEET : ZC_DRY, ckpt, lane, bits)) fin = st;
        }
    }
    __builtin_amdgcn_wave_barrier();                                          // all reads of pass 1 are done: records may replace codes
    // pass 2: every slice again from its (now true) entering state, records written in place
    uint32_t st = enter;
    fse_chain_run(ct, arr, M, topC, botC, st, ZC_RECORD, ckpt, lane, bits);
    if (recBits) *recBits = tw_sum(bits);                                     // G = 21: of the three chains together
    return __shfl(fin, (int)(grp * G + used - 1));                            // the group's lowest slice holds the final state
}

// ZSTD_fseBitCost (zstd_compress_sequences.c:103-135) with FSE_bitCost (lib/common/fse.h:494-509): the cost in bits of coding the
// histogram (lane = symbol, myCnt = its count, max = the largest symbol present) with a previous table; ~0 when a symbol that occurs
// has no cell in it (its transform is the zero-probability one: the "bad" cost).  Wave-wide, the same value in every lane.
__device__ inline uint64_t fse_bit_cost_wave(const FseCTable* __restrict__ ct, uint32_t myCnt, uint32_t max)
{
    uint32_t const lane = tw_lane();
    uint32_t const tableLog = ct->tableLog, badCost = (tableLog + 1) << 8;
    uint32_t cost = 0; bool bad = false;
    if (lane <= max && lane < 56 && myCnt != 0) {
        uint32_t const dBits = ct->dBits[lane];
        uint32_t const minNbBits = dBits >> 16, threshold = (minNbBits + 1) << 16;
        uint32_t const deltaFromThreshold = threshold - (dBits + (1u << tableLog));
        uint32_t const normalized = (deltaFromThreshold << 8) >> tableLog;
        uint32_t const bitCost = (minNbBits + 1) * 256 - normalized;
        bad = bitCost >= badCost;
        cost = myCnt * bitCost;
    }
    if (__ballot(bad)) return ~0ull;
    return (uint64_t)(tw_sum(cost) >> 8);
}

// ================================================================== the block encoder
// Literals section + sequences section of one block (ZSTD_entropyCompressSeqStore, zstd_compress.c:3000-3050) into body[];
// returns the compressed size, or 0 when the block has to be emitted uncompressed (same value in every thread).  n >= 7.
// `de`: the previous block's / the dictionary's entropy state, or nullptr.  Afterwards sh->litMode / litType / code[] /
// hufMaxSym describe the literals' Huffman table (a multi-block frame keeps it as the next block's previous table).
// NT threads per workgroup: 256 (four wavefronts: the literals job and the three sequence tables side by side, one Huffman stream
// per wavefront) with SH = EntShared, or 64 (one wavefront does the jobs one after the other) with SH = EntSharedSmall — the form for
// small records, where a 256-thread workgroup would mostly wait at its own barriers.
template <uint32_t NT, typename SH>
__device__ inline uint32_t entropy_block(const uint8_t* __restrict__ src, uint32_t n, const ZhipUnit& u, const ZhipSeq* __restrict__ seqs,
                                         const ZhipParse& pm, const uint8_t* __restrict__ lits, uint16_t* __restrict__ stBits, uint32_t seqCap,
                                         uint8_t* __restrict__ body, SH* sh, const ZhipDictEntropy* __restrict__ de)
{
    constexpr uint32_t NW = NT / 64;                         // wavefronts of the workgroup
    constexpr uint32_t SPW = 4 / NW;                         // Huffman streams a wavefront takes care of
    int const t = (int)threadIdx.x, lane = t & 63, wv = t >> 6;
    uint32_t const nbSeq = pm.nbSeq;
    uint32_t const minGainBlock = (n >> 6) + 2;             // zstd_compress_internal.h:613
    ZPROF_DECL
    uint16_t* const bLL = stBits; uint16_t* const bOF = stBits + seqCap; uint16_t* const bML = stBits + 2 * (size_t)seqCap;

    // ================ phase A (all threads): byte histogram of the literals; sequence codes + code histograms
    for (int i = t; i < (int)(NW * 256); i += NT) (&sh->hist[0][0])[i] = 0;
    for (int i = t; i < 3 * 64; i += NT) (&sh->seqCount[0][0])[i] = 0;
    if (t < 64) sh->tabs.llCode[t] = kLLcode[t];
    for (int i = t; i < 128; i += NT) sh->tabs.mlCode[i] = kMLcode[i];
    if (t < 36) sh->tabs.llBits[t] = kLLbits[t];
    if (t < 53) sh->tabs.mlBits[t] = kMLbits[t];
    const CodeTabs& TB = sh->tabs;
    __syncthreads();
    // byte histogram of the literals the match finder left in lits[] (HIST_count_wksp, hist.c:154): 16 bytes per thread
    // per step, coalesced; one histogram per wavefront to spread the LDS atomics
    uint32_t const litSize = pm.litSize;
#ifndef ZHIP_HUF_HISTSIZE
#define ZHIP_HUF_HISTSIZE 1
#endif
    constexpr bool histSized = (NT == 256) && (ZHIP_HUF_HISTSIZE != 0);
    {   uint32_t* const myHist = sh->hist[wv];
        // histSized (round 6): with four streams (litSize >= 256, zstd_compress_literals.c:142 — a valid previous table's single stream is handled below)
        // wavefront w takes segment w of huf_compress.c:1168-1215, so hist[w] is the histogram of stream w
        bool const perStream = histSized && litSize >= 256;
        uint32_t const segA = (litSize + 3) / 4;
        uint32_t const lo = perStream ? (uint32_t)wv * segA : 0u;
        uint32_t const hi = perStream ? ((uint32_t)wv < 3 ? lo + segA : litSize) : litSize;
        uint32_t const tA = perStream ? (uint32_t)lane : (uint32_t)t;
        uint32_t const strideB = perStream ? 16u * 64u : 16u * NT;
        for (uint32_t i0 = lo + 16u * tA; i0 < hi; i0 += 4 * strideB) {      // 4 loads in flight per thread
            uint4 v[4];
            for (int q = 0; q < 4; q++) { uint32_t const i = i0 + (uint32_t)q * strideB; if (i < hi) __builtin_memcpy(&v[q], lits + i, 16); }   // lits has >= 64 bytes of slack
            for (int q = 0; q < 4; q++) {
                uint32_t const i = i0 + (uint32_t)q * strideB;
                if (i >= hi) break;
                uint32_t const c = hi - i < 16 ? hi - i : 16;
                uint32_t const w[4] = { v[q].x, v[q].y, v[q].z, v[q].w };
                for (uint32_t b = 0; b < 16; b++) if (b < c) atomicAdd(&myHist[(w[b >> 2] >> (8 * (b & 3))) & 0xFF], 1u);
            }
        }
    }
    ZPROF(9);
    // sequence codes (zstd_compress.c:2686-2712) -> stBits arrays (replaced by the FSE records in phase B) + histograms
    for (uint32_t i = (uint32_t)t; i < nbSeq; i += NT) {
        uint32_t ll, mlb, ob; seq_fields(seqs, pm, i, ll, mlb, ob);
        uint32_t const llc = ll_code(TB, ll), ofc = hb32(ob), mlc = ml_code(TB, mlb);
        bLL[i] = (uint16_t)llc; bOF[i] = (uint16_t)ofc; bML[i] = (uint16_t)mlc;
        atomicAdd(&sh->seqCount[0][llc], 1u);
        atomicAdd(&sh->seqCount[1][ofc], 1u);
        atomicAdd(&sh->seqCount[2][mlc], 1u);
    }
    // literals-section geometry and the sampling heuristic's two histograms (huf_compress.c:1367-1379)
    ZPROF(10);
    uint8_t* const litDst = body;
    uint32_t const lhSize = 3 + (litSize >= 1024) + (litSize >= 16384);
    uint32_t const hufRep0 = de ? de->hufRepeat : 0;        // previous Huffman table: 0 none, 1 check, 2 valid (dictionary)
    bool const single = litSize < 256 || (hufRep0 == 2 && lhSize == 3);                // zstd_compress_literals.c:142, :170
#ifdef ZHIP_PROBE_NOHUF             /* timing probe, bytes wrong */
    bool const tryHuf0 = false;
#else
    bool const tryHuf0 = !(u.litMode) && litSize >= (hufRep0 == 2 ? 6u : 64u);
#endif        // :115-127 minLiteralsToCompress (strategy <= lazy2)
    bool const suspect = (nbSeq == 0) || (litSize / nbSeq >= 20);                      // zstd_compress.c:2918
    bool const sampling = tryHuf0 && suspect && litSize >= 40960;
    if (sampling) for (int i = t; i < 512; i += NT) (&sh->sampleHist[0][0])[i] = 0;
    __syncthreads();                                        // lits[] complete, histograms complete
    if (NW > 1) for (int i = t; i < 256; i += NT) { uint32_t a = 0; for (uint32_t w = 1; w < NW; w++) a += sh->hist[w][i]; sh->hist[0][i] += a; }
    if (sampling) {
        for (uint32_t i = (uint32_t)t; i < 4096; i += NT) {
            atomicAdd(&sh->sampleHist[0][lits[i]], 1u);
            atomicAdd(&sh->sampleHist[1][lits[litSize - 4096 + i]], 1u);
        }
    }
    __syncthreads();
    ZPROF(0);

    // ================ phase B (four wavefronts, concurrently): wave 0 decides the literals mode and builds the Huffman code,
    // waves 1..3 select and build one sequence table each (LL, OF, ML) and then run that table's state chain.  The table
    // builders of zhip_tables.h are wave-wide: one lane per symbol, all decisions wave-uniform.
    if (wv == 0) {
        // literals (zstd_compress_literals.c:129-235 with no previous table)
        ZPROF_JOB_BEGIN
        bool tryHuf = tryHuf0;
        if (sampling) {
            uint32_t a = 0, b = 0;
            for (int s0 = lane; s0 < 256; s0 += 64) { if (sh->sampleHist[0][s0] > a) a = sh->sampleHist[0][s0]; if (sh->sampleHist[1][s0] > b) b = sh->sampleHist[1][s0]; }
            a = tw_max(a); b = tw_max(b);
            if (a + b <= ((2 * 4096) >> 7) + 4) tryHuf = false;
        }
        uint32_t mode = 0, hdrSize = 0, litType = 2, logOut = 0;      // raw
        // HUF_compress_internal (huf_compress.c:1333-1434) with the dictionary's table as the previous one when there is one
        uint32_t rep = hufRep0;
        bool const preferRepeat = u.strategy < ZHIP_STRAT_LAZY && litSize <= 1024;   // zstd_compress_literals.c:165
        bool useOld = false;
        if (tryHuf0 && preferRepeat && rep == 2) { mode = 2; useOld = true; }        // :1359-1363 valid table, small input: no statistics at all
        else if (tryHuf) {
            uint32_t hq[4], big = 0, topSym = 0;
            for (int q = 0; q < 4; q++) { hq[q] = sh->hist[0][lane + 64 * q]; if (hq[q] > big) big = hq[q]; if (hq[q]) topSym = (uint32_t)(lane + 64 * q) + 1; }
            uint32_t const maxSym = tw_max(topSym) - 1, largest = tw_max(big);
            if (largest == litSize) mode = 1;                                        // huf_compress.c:1383 -> RLE literals
            else if (largest <= (litSize >> 7) + 4) mode = 0;                        // :1384
            else {
                if (rep == 1) {                                                      // :1389-1393 HUF_validateCTable
                    bool miss = false;
                    for (int q = 0; q < 4; q++) { uint32_t const sy = (uint32_t)(lane + 64 * q); miss = miss || (sy <= maxSym && hq[q] != 0 && (de->hufCode[sy] & 0xFF) == 0); }
                    if (de->hufMaxSym < maxSym || __ballot(miss)) rep = 0;
                }
                if (preferRepeat && rep != 0) { mode = 2; useOld = true; }           // :1395-1399
                else {
                    uint32_t huffLog = fse_optimal_table_log(11, litSize, maxSym, 1);    // :1284-1287
                    ZPROF_JOB_MARK(31);
                    huffLog = huf_build_codes_wave(&sh->huf, sh->hist[0], maxSym, huffLog, sh->code);
                    ZPROF_JOB_MARK(28);
                    uint32_t const h = huf_write_table_wave(&sh->huf, sh->hufHdr, sh->code, maxSym, huffLog);
                    if (h != 0 && rep != 0) {                                        // :1415-1421 is the previous table cheaper?
                        uint32_t oldSize = 0, newSize = 0;
                        for (int q = 0; q < 4; q++) { uint32_t const sy = (uint32_t)(lane + 64 * q); if (sy <= maxSym) { oldSize += (de->hufCode[sy] & 0xFF) * hq[q]; newSize += (sh->code[sy] & 0xFF) * hq[q]; } }
                        oldSize = tw_sum(oldSize); newSize = tw_sum(newSize);
                        if ((oldSize >> 3) <= h + (newSize >> 3) || h + 12 >= litSize) { mode = 2; useOld = true; }
                    }
                    if (!useOld && h != 0 && h + 12 < litSize) { mode = 2; hdrSize = h; logOut = huffLog; if (lane == 0) sh->hufMaxSym = maxSym; }   // :1425
                }
            }
        }
        if (useOld) { hdrSize = 0; litType = 3; }                            // set_repeat: treeless literals, the dictionary's code
        if (lane == 0) { sh->hufHdrSize = hdrSize; sh->litType = litType; sh->huffLog = logOut; sh->litMode = mode; }
        __builtin_amdgcn_wave_barrier();
        if (de && mode == 2 && litType == 3) for (int s0 = lane; s0 < 256; s0 += 64) sh->code[s0] = de->hufCode[s0];
        ZPROF_JOB_MARK(31);
    }
    if (NW == 1) __builtin_amdgcn_wave_barrier();          // one wavefront: the literals job is done with its workspace before the tables reuse it
    if ((NW == 1 || wv >= 1) && nbSeq > 0)
    for (int k = (NW == 1 ? 0 : wv - 1); k < (NW == 1 ? 3 : wv); k++) {                                  // 0 LL, 1 OF, 2 ML
        uint16_t* const arr = stBits + (size_t)k * seqCap;
        uint32_t const lastCode = arr[nbSeq - 1];                  // for the "-1" rule (zstd_compress_sequences.c:271-274)
        ZPROF_JOB_BEGIN
        {
            uint32_t const maxPossible = (k == 0) ? 35 : (k == 1 ? 31 : 52);
            uint32_t const defLog = (k == 1) ? 5 : 6, fseLog = (k == 1) ? 8 : 9;
            uint32_t const defMax = (k == 0) ? 35 : (k == 1 ? 28 : 52);
            const int16_t* defNorm = (k == 0) ? kLLnorm : (k == 1 ? kOFnorm : kMLnorm);
            uint32_t* cnt = sh->seqCount[k];
            uint32_t const myCnt = (uint32_t)lane <= maxPossible ? cnt[lane] : 0;        // lane = code
            {   // the extra bits of all sequences in this field (zstd_internal.h:123-168; an offset code is its own number of extra bits), before the "-1" rule touches the counts
                uint32_t const eb = (uint32_t)lane <= maxPossible ? (k == 0 ? TB.llBits[lane] : (k == 1 ? (uint32_t)lane : TB.mlBits[lane])) : 0u;
                uint32_t const ext = tw_sum(myCnt * eb);
                if (lane == 0) sh->extraBits[k] = ext;
            }
            uint32_t const max = 63u - (uint32_t)__clzll((long long)__ballot(myCnt != 0));
            uint32_t const mostFrequent = tw_max(myCnt);
            bool const defaultAllowed = (k != 1) || (max <= 28);                       // zstd_compress.c:2814
            // ZSTD_selectEncodingType (zstd_compress_sequences.c:157-235)
            uint32_t type;
            uint32_t hsz = 0; bool fail = false;
            if (mostFrequent == nbSeq) type = (defaultAllowed && nbSeq <= 2) ? 0 : 1;
            else if (u.strategy < ZHIP_STRAT_LAZY) {
                type = 2;
                if (defaultAllowed) {
                    uint32_t const mult = 10 - u.strategy;
                    uint32_t const dynMin = ((1u << defLog) * mult) >> 3;
                    if (de && de->fseRepeat[k] == 2 && nbSeq < 1000) type = 3;        // :187-191 set_repeat: the dictionary's VALID table
                    else if (nbSeq < dynMin || mostFrequent < (nbSeq >> (defLog - 1))) type = 0;
                }
            } else {
                // strategy >= lazy (:205-231): estimated costs in bits; without a previous table the repeat cost is an
                // error (= larger than everything)
                uint64_t basicCost = ~0ull;
                if (defaultAllowed) {                                                  // :141-155 ZSTD_crossEntropyCost
                    uint32_t const shift = 8 - defLog;
                    uint32_t mine = 0;
                    if ((uint32_t)lane <= max) { uint32_t const normAcc = defNorm[lane] != -1 ? (uint32_t)defNorm[lane] : 1; mine = myCnt * kInvProbLog256[normAcc << shift]; }
                    basicCost = tw_sum(mine) >> 8;
                }
                uint32_t const tl = fse_optimal_table_log(fseLog, nbSeq, max, 2);        // :70-77 ZSTD_NCountCost
                uint32_t ncountCost = 0;
                if (fse_normalize_wave(sh->norm[k], tl, cnt, nbSeq, max, nbSeq >= 2048) < 0) fail = true;
                else { ncountCost = fse_write_ncount_wave(sh->ncWords[k], sh->ncount[k], sh->norm[k], max, tl); if (!ncountCost) fail = true; }
                uint32_t mineE = 0;                                                    // :83-97 ZSTD_entropyCost
                if ((uint32_t)lane <= max) {
                    uint32_t nrm = (256 * myCnt) / nbSeq;
                    if (myCnt != 0 && nrm == 0) nrm = 1;
                    mineE = myCnt * kInvProbLog256[nrm];
                }
                uint32_t const ecost = tw_sum(mineE);
                uint64_t const compressedCost = ((uint64_t)ncountCost << 3) + (ecost >> 8);
                // :212 the previous block's table (multi-block frames of the lazy strategies, zhip_frame_lazy.h): ZSTD_fseBitCost
                uint64_t const repeatCost = (de && de->fseRepeat[k] != 0) ? fse_bit_cost_wave(&de->ct[k], myCnt, max) : ~0ull;
                if (basicCost <= repeatCost && basicCost <= compressedCost) type = 0;  // :217-222
                else if (repeatCost <= compressedCost) type = 3;                       // :223-227 set_repeat, the table's state stays
                else type = 2;
            }
            if (type == 3) {           // zstd_compress_sequences.c:264-266: the previous table as it is, no header bytes
                                       // (copied below by the whole wavefront)
            } else if (type == 1) {    // set_rle: the single symbol is `max`; byte = code of the first sequence (= same)
                fse_build_ctable_rle_wave(&sh->ct[k], max);
                if (lane == 0) sh->ncount[k][0] = (uint8_t)max;
                hsz = 1;
            } else if (type == 0) {
                if ((uint32_t)lane <= defMax) sh->norm[k][lane] = defNorm[lane];
                __builtin_amdgcn_wave_barrier();
                fse_build_ctable_wave(&sh->ct[k], sh->norm[k], defMax, defLog, sh->symScratch[NW == 1 ? 0 : k], sh->cumul[k]);
            } else {
                uint32_t nbSeq1 = nbSeq;
                uint32_t const tableLog = fse_optimal_table_log(fseLog, nbSeq, max, 2);
                if (__builtin_amdgcn_readlane(myCnt, (int)lastCode) > 1) { if ((uint32_t)lane == lastCode) cnt[lastCode]--; nbSeq1--; }
                __builtin_amdgcn_wave_barrier();
                if (fse_normalize_wave(sh->norm[k], tableLog, cnt, nbSeq1, max, nbSeq1 >= 2048) < 0) fail = true;
                else {
                    hsz = fse_write_ncount_wave(sh->ncWords[k], sh->ncount[k], sh->norm[k], max, tableLog);
                    if (!hsz) fail = true;
                    else fse_build_ctable_wave(&sh->ct[k], sh->norm[k], max, tableLog, sh->symScratch[NW == 1 ? 0 : k], sh->cumul[k]);
                }
            }
            if (lane == 0) { sh->encType[k] = fail ? 9 : type; sh->ncountSize[k] = hsz; sh->maxCode[k] = max; }
        }
        __builtin_amdgcn_wave_barrier();
        if (sh->encType[k] == 3) {
            const uint32_t* const from = (const uint32_t*)&de->ct[k];
            uint32_t* const to = (uint32_t*)&sh->ct[k];
            for (uint32_t i = (uint32_t)lane; i < sizeof(FseCTable) / 4; i += 64) to[i] = from[i];
            __builtin_amdgcn_wave_barrier();
        }
        ZPROF_JOB_MARK(29);
        if (NW > 1 && sh->encType[k] != 9) {
            // the table's FSE state chain, last sequence -> first (zstd_compress_sequences.c:311-369): arr[i] (the code)
            // becomes (nbBits << 12 | value)
            // scratch for the slices' chunk states: the block's own output room, which nothing has written yet (3 x <= 8 KB of it)
            uint16_t* const ckpt = (uint16_t*)(((uintptr_t)body + 15) & ~(uintptr_t)15) + (size_t)k * fse_chain_ckpt_entries(nbSeq - 1);
#ifdef ZHIP_PROBE_NOCHAIN          /* timing probe, bytes wrong */
            uint32_t const fin = lastCode; (void)ckpt;
#else
            uint32_t cb = 0;
            uint32_t const fin = fse_chain_wave<64>(&sh->ct[k], arr, nbSeq - 1, lastCode, ckpt, &cb);
            if (lane == 0) sh->chainBits[k] = cb;
#endif
            if (lane == 0) sh->finalState[k] = fin;
        }
        ZPROF_JOB_MARK(30);
    }
    if (NW == 1 && nbSeq > 0) {
        // one wavefront: the three chains side by side, 21 lanes each (a small block's chains are a handful of dependent round
        // trips each — three in a row would be most of the kernel)
        __builtin_amdgcn_wave_barrier();
        bool const anyFail = (sh->encType[0] == 9) | (sh->encType[1] == 9) | (sh->encType[2] == 9);   // never for valid histograms; the block goes raw
        if (!anyFail) {
            uint32_t const g = (uint32_t)lane / 21u, k3 = g < 3 ? g : 2;
            uint16_t* const arr3 = stBits + (size_t)k3 * seqCap;
            uint16_t* const ckpt = (uint16_t*)(((uintptr_t)body + 15) & ~(uintptr_t)15);
#ifdef ZHIP_PROBE_NOCHAIN          /* timing probe, bytes wrong */
            uint32_t const fin = arr3[nbSeq - 1]; (void)ckpt;
#else
            uint32_t cb = 0;
            uint32_t const fin = fse_chain_wave<21>(&sh->ct[k3], arr3, nbSeq - 1, arr3[nbSeq - 1], ckpt, &cb);
            if (lane == 0) { sh->chainBits[0] = cb; sh->chainBits[1] = 0; sh->chainBits[2] = 0; }
#endif
            if (lane == 0 || lane == 21 || lane == 42) sh->finalState[k3] = fin;
        }
    }
    __syncthreads();
    ZPROF(1);

    // ================ phase C: literals section
    uint32_t litMode = sh->litMode;
    if (litMode == 2) {
        // stream geometry (huf_compress.c:1168-1215): 4 segments of (litSize+3)/4, or one stream
        uint32_t const nStreams = single ? 1 : 4;
        uint32_t const seg = single ? litSize : (litSize + 3) / 4;
        // Two forms of the literal pack, the same bits: small units (one wavefront per unit, NT == 64: records of a kilobyte) walk a stream in chunks of
        // 64 x 16 symbols, lane l taking the 16 bytes at chunk + 16 l — coalesced (records leg: entropy 108 -> 96 ms per 10 M records); full-size units keep
        // one contiguous run of the stream per lane, whose lines stay in L1 from round to round (the chunked form was 10 % slower there: A/B in
        // profiles/r04_entropy_tiles2.log)
        constexpr bool chunked = NT == 64;
        // Round 6: full-size units pack through LDS (ZHIP_HUF_STAGE).  A wavefront walks its stream in chunks of 64 x 16 symbols (one coalesced kilobyte per load),
        // the lanes OR their codes into the wavefront's own LDS image of the chunk's stretch of the stream, and the image leaves with whole-word stores, 256
        // contiguous bytes per instruction; the chunk's last partial word stays behind as the next image's first.  Every byte of a stream is stored exactly once
        // and only by its own wavefront (the stream's first and last words go out byte by byte where they are shared), so the zeroing pass, its barrier and the
        // global atomics are gone.  The per-lane form it replaces touched 64 lines per store instruction, 16 bytes of each (WRITE_SIZE 4.7 x the frames).
#ifndef ZHIP_HUF_STAGE
#define ZHIP_HUF_STAGE 1
#endif
#ifndef ZHIP_HUF_STAGE_SMALL
#define ZHIP_HUF_STAGE_SMALL 1
#endif
        // (the one-wavefront form stages too: its image is its one histogram, 256 words, dead once the stream sizes are known — chunks of 64 x 8 symbols there,
        // 31 + 512 x 11 bits at most; a single stream — every record that codes its literals with the dictionary's table — is sized from the block's histogram, four
        // streams by the chunked pass)
        constexpr bool staged = (NT == 256) ? (ZHIP_HUF_STAGE != 0) : (ZHIP_HUF_STAGE_SMALL != 0);
        constexpr bool sizeChunked = chunked || (staged && ZHIP_HUF_STAGE == 2);
        uint32_t segStart[SPW], segLenA[SPW], total[SPW], runStart[SPW], runLen[SPW], incl[SPW];
        if constexpr (NT == 256 && staged && histSized) {
            // no pass 1 (round 6): stream w's bits = sum over the symbols of hist[w][s] x nbBits(s); hist[0] holds the block's histogram by now, so stream 0 is
            // the block minus the other three.  Every wavefront computes its own stream's; a single stream is the block's.
            uint32_t const sI = (uint32_t)wv;
            segStart[0] = 0; segLenA[0] = 0; total[0] = 0;
            if (sI < nStreams) {
                segStart[0] = sI * seg;
                segLenA[0] = single ? litSize : ((sI < 3) ? seg : litSize - 3 * seg);
                uint32_t mine = 0;
#pragma unroll
                for (uint32_t q = 0; q < 4; q++) {
                    uint32_t const sy = (uint32_t)lane + 64u * q;
                    uint32_t cnt = sh->hist[sI][sy];
                    if (sI == 0 && !single) cnt -= sh->hist[1][sy] + sh->hist[2][sy] + sh->hist[3][sy];
                    mine += cnt * (sh->code[sy] & 0xFF);
                }
                uint32_t const inclAll = wave_incl_scan(mine);
                total[0] = __shfl(inclAll, 63);
                if (lane == 0) { sh->streamBits[sI] = total[0]; sh->streamBytes[sI] = (total[0] >> 3) + 1; }
            }
        } else if constexpr (sizeChunked) {
            // pass 1: wavefront `wv` sizes stream `wv`.  Round 4: COALESCED — the wavefront walks the stream in chunks of 64 x 16 symbols, lane l takes
            // the 16 bytes at chunk + 16 l (one 1 KB stretch per load instruction; a lane used to own one contiguous run of the stream, i.e. 64 lines per
            // load).  (lits[] has ZHIP_LIT_STRIDE - ZHIP_UNIT_MAX bytes of slack, so a 16-byte read may run past the stream.)
#pragma unroll
            for (uint32_t q = 0; q < SPW; q++) {
                uint32_t const sI = (uint32_t)wv + q * NW;                      // stream of this wavefront's q-th turn
                uint32_t myBits = 0;
                segStart[q] = 0; segLenA[q] = 0;
                if (NT == 64 && staged && single) {
                    // one stream = the block: its bits are sum(count x code length) over the block's histogram, no pass over the literals
                    if (sI == 0) {
                        segLenA[q] = litSize;
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++) myBits += sh->hist[0][lane + 64u * k] * (sh->code[lane + 64u * k] & 0xFF);
                    }
                } else
                if (sI < nStreams) {
                    segStart[q] = sI * seg;
                    uint32_t const segLen = single ? litSize : ((sI < 3) ? seg : litSize - 3 * seg);
                    segLenA[q] = segLen;
                    const uint8_t* const p = lits + segStart[q];
                    for (uint32_t c0 = 16u * (uint32_t)lane; c0 < segLen; c0 += 1024u) {
                        uint4 va; __builtin_memcpy(&va, p + c0, 16);
                        uint32_t const w[4] = { va.x, va.y, va.z, va.w };
                        uint32_t const c = segLen - c0;
#pragma unroll
                        for (uint32_t b = 0; b < 16; b++) if (b < c) myBits += sh->code[(w[b >> 2] >> (8 * (b & 3))) & 0xFF] & 0xFF;
                    }
                }
                uint32_t const inclAll = wave_incl_scan(myBits);
                total[q] = __shfl(inclAll, 63);
                if (lane == 0 && sI < nStreams) { sh->streamBits[sI] = total[q]; sh->streamBytes[sI] = (total[q] >> 3) + 1; }
            }
        } else {
            // pass 1: wavefront `wv` sizes stream `wv`.  lane owns a contiguous run of symbols, read 8 at a time
            // (lits[] has ZHIP_LIT_STRIDE - ZHIP_UNIT_MAX bytes of slack, so an 8-byte read may run past the run).
#pragma unroll
            for (uint32_t q = 0; q < SPW; q++) {
                uint32_t const sI = (uint32_t)wv + q * NW;                      // stream of this wavefront's q-th turn
                uint32_t myBits = 0;
                segStart[q] = 0; runStart[q] = 0; runLen[q] = 0; segLenA[q] = 0;
                if (sI < nStreams) {
                    segStart[q] = sI * seg;
                    uint32_t const segLen = single ? litSize : ((sI < 3) ? seg : litSize - 3 * seg);
                    segLenA[q] = segLen;
                    uint32_t const rper = (segLen + 63) / 64;
                    runStart[q] = (uint32_t)lane * rper; if (runStart[q] > segLen) runStart[q] = segLen;
                    runLen[q] = (runStart[q] + rper <= segLen) ? rper : segLen - runStart[q];
                    const uint8_t* p = lits + segStart[q] + runStart[q];
                    for (uint32_t i = 0; i < runLen[q]; i += 32) {
                        uint4 va, vb; __builtin_memcpy(&va, p + i, 16); __builtin_memcpy(&vb, p + i + 16, 16);
                        uint32_t const w[8] = { va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w };
                        uint32_t const c = runLen[q] - i;
                        for (uint32_t b = 0; b < 32; b++) if (b < c) myBits += sh->code[(w[b >> 2] >> (8 * (b & 3))) & 0xFF] & 0xFF;
                    }
                }
                incl[q] = wave_incl_scan(myBits);
                total[q] = __shfl(incl[q], 63);
                if (lane == 0 && sI < nStreams) { sh->streamBits[sI] = total[q]; sh->streamBytes[sI] = (total[q] >> 3) + 1; }
            }
        }
        __syncthreads();
        if (t == 0) {
            uint32_t off = lhSize + sh->hufHdrSize + (single ? 0 : 6), tot = sh->hufHdrSize + (single ? 0 : 6);
            bool ok = true;
            for (uint32_t k = 0; k < nStreams; k++) { sh->streamOff[k] = off; off += sh->streamBytes[k]; tot += sh->streamBytes[k]; if (sh->streamBytes[k] > 65535) ok = false; }
            // huf_compress.c:1237 (>= srcSize-1 -> 0) then zstd_compress_literals.c:187-191 (minGain)
            if (!ok || tot >= litSize - 1 || tot >= litSize - ((litSize >> 6) + 2)) sh->litMode = 0;
            else if (tot == 1 && litSize < 8) {           // zstd_compress_literals.c:192-201: a 1-byte result of identical bytes is RLE
                bool same = true;
                for (uint32_t i = 1; i < litSize; i++) same = same && lits[i] == lits[0];
                if (same) sh->litMode = 1;
            }
            sh->litSectionSize = lhSize + tot;
        }
        __syncthreads();
        litMode = sh->litMode;
        if (litMode == 2) {
            uint32_t const cLit = sh->litSectionSize - lhSize;
            // zero the stream bytes that will be OR-ed (the staged packer stores every byte of a stream itself: no zeroing, no shared words)
            if constexpr (!staged) zero_bytes<NT>(litDst + sh->streamOff[0], sh->litSectionSize - sh->streamOff[0]);
            if (t == 0) {
                // section header (zstd_compress_literals.c:209-232)
                if (lhSize == 3) { uint32_t const lhc = sh->litType + ((uint32_t)(!single) << 2) + (litSize << 4) + (cLit << 14); litDst[0] = (uint8_t)lhc; litDst[1] = (uint8_t)(lhc >> 8); litDst[2] = (uint8_t)(lhc >> 16); }
                else if (lhSize == 4) { uint32_t const lhc = sh->litType + (2 << 2) + (litSize << 4) + (cLit << 18); litDst[0] = (uint8_t)lhc; litDst[1] = (uint8_t)(lhc >> 8); litDst[2] = (uint8_t)(lhc >> 16); litDst[3] = (uint8_t)(lhc >> 24); }
                else { uint32_t const lhc = sh->litType + (3 << 2) + (litSize << 4) + (cLit << 22); litDst[0] = (uint8_t)lhc; litDst[1] = (uint8_t)(lhc >> 8); litDst[2] = (uint8_t)(lhc >> 16); litDst[3] = (uint8_t)(lhc >> 24); litDst[4] = (uint8_t)(cLit >> 10); }
                for (uint32_t i = 0; i < sh->hufHdrSize; i++) litDst[lhSize + i] = sh->hufHdr[i];
                if (!single) for (int k = 0; k < 3; k++) { uint8_t* jt = litDst + lhSize + sh->hufHdrSize + 2 * k; jt[0] = (uint8_t)sh->streamBytes[k]; jt[1] = (uint8_t)(sh->streamBytes[k] >> 8); }
            }
            if constexpr (!staged) __syncthreads();      // zeroing + header byte stores are complete before any atomicOr touches a shared word
            ZPROF(2);
            // pass 2: pack.  Symbols are emitted last -> first (huf_compress.c:1056-1118): the bit position of a run
            // is the number of bits of all LATER symbols of the stream = total - inclusive prefix.
            if constexpr (staged) {
#pragma unroll 1
                for (uint32_t q = 0; q < SPW; q++) {
                uint32_t const sI = (uint32_t)wv + q * NW;                         // four wavefronts: wavefront = stream; one wavefront: the streams one after the other
                if (sI < nStreams) {
                    uint8_t* const sbase = litDst + sh->streamOff[sI];
                    uint32_t* const w32 = (uint32_t*)((uintptr_t)sbase & ~(uintptr_t)3);
                    uint8_t* const w8 = (uint8_t*)w32;
                    uint32_t const firstB = (uint32_t)((uintptr_t)sbase & 3);      // the stream's bytes are [firstB, endB) counted from w32
                    uint32_t const endB = firstB + sh->streamBytes[sI];
                    // 1 536 bytes per wavefront (31 + 1 024 x 11 bits at most): hist[] + sampleHist[], dead since phase B; the one-wavefront form: its histogram, 8 symbols per lane
                    constexpr uint32_t SYM = NT == 256 ? 16u : 8u, CHLOG = NT == 256 ? 10u : 9u, IMGW = NT == 256 ? 384u : 256u;
                    uint32_t* const img = NT == 256 ? &sh->hist[0][0] + 384u * (uint32_t)wv : &sh->hist[0][0];
                    const uint8_t* const p = lits + segStart[q];
                    uint32_t const segLen = segLenA[q];
                    uint32_t const nChunks = (segLen + (1u << CHLOG) - 1u) >> CHLOG;   // wave-uniform (stream geometry)
                    for (uint32_t w = (uint32_t)lane; w < IMGW; w += 64u) img[w] = 0;
                    __threadfence_block();
                    __builtin_amdgcn_wave_barrier();
                    uint32_t carry = 8u * firstB;                                   // bits of image word 0 in front of this chunk's
                    uint32_t wordBase = 0;                                          // image word 0 is w32[wordBase]
                    auto flush = [&](uint32_t words) {                              // image words [0, words) -> the frame; words shared with a neighbour section go byte by byte
                        for (uint32_t w = (uint32_t)lane; w < words; w += 64u) {
                            uint32_t const v = img[w], g = wordBase + w, b0 = 4u * g;
                            if (b0 >= firstB && b0 + 4u <= endB) w32[g] = v;
                            else for (uint32_t k = 0; k < 4; k++) if (b0 + k >= firstB && b0 + k < endB) w8[b0 + k] = (uint8_t)(v >> (8 * k));
                        }
                    };
                    for (uint32_t ch = nChunks; ch-- > 0; ) {
                        // chunks from the stream's end to its start; inside a chunk the higher lanes' symbols come first
                        uint32_t const c0 = (ch << CHLOG) + SYM * (uint32_t)lane;
                        uint32_t const c = c0 < segLen ? (segLen - c0 < SYM ? segLen - c0 : SYM) : 0u;
                        uint32_t cd[SYM]; uint32_t myBits = 0;
                        if (c) {
                            uint32_t w[SYM / 4]; __builtin_memcpy(w, p + c0, SYM);
#pragma unroll
                            for (uint32_t b = 0; b < SYM; b++) { cd[b] = b < c ? sh->code[(w[b >> 2] >> (8 * (b & 3))) & 0xFF] : 0u; myBits += cd[b] & 0xFF; }
                        }
                        uint32_t const inclC = wave_incl_scan(myBits);
                        uint32_t const chunkBits = __shfl(inclC, 63);
                        if (c) {
                            uint32_t const pos = carry + chunkBits - inclC;
                            uint32_t* wp = img + (pos >> 5); uint64_t acc = 0; uint32_t have = pos & 31;
#pragma unroll
                            for (int b = (int)SYM - 1; b >= 1; b -= 2) {        // two symbols (<= 22 bits) per step, last -> first
                                uint32_t const c1 = cd[b], c0v = cd[b - 1];
                                uint32_t const n1 = c1 & 0xFF;
                                acc |= (uint64_t)((c1 >> 8) | ((c0v >> 8) << n1)) << have; have += n1 + (c0v & 0xFF);
                                if (have >= 32) { __hip_atomic_fetch_or(wp, (uint32_t)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); wp++; acc >>= 32; have -= 32; }
                            }
                            if (have) __hip_atomic_fetch_or(wp, (uint32_t)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        }
                        __threadfence_block();
                        __builtin_amdgcn_wave_barrier();
                        uint32_t const tot = carry + chunkBits, full = tot >> 5;
                        flush(full);
                        uint32_t const cw = img[full];                              // the partial word: first of the next image
                        __threadfence_block();
                        __builtin_amdgcn_wave_barrier();
                        for (uint32_t w = (uint32_t)lane; w <= full; w += 64u) img[w] = w ? 0u : cw;
                        __threadfence_block();
                        __builtin_amdgcn_wave_barrier();
                        carry = tot & 31; wordBase += full;
                    }
                    // the stream's FIRST symbols came last: the end mark follows them; what is left is at most one word
                    if (lane == 0) img[0] |= 1u << carry;
                    __threadfence_block();
                    __builtin_amdgcn_wave_barrier();
                    flush(1);
                    __threadfence_block();
                    __builtin_amdgcn_wave_barrier();
                }
                }
            } else if constexpr (chunked) {
#pragma unroll
                for (uint32_t q = 0; q < SPW; q++) {
                    uint32_t const sI = (uint32_t)wv + q * NW;
                    if (sI >= nStreams) continue;
                    uint8_t* const sbase = litDst + sh->streamOff[sI];
                    uint32_t* const w32 = (uint32_t*)((uintptr_t)sbase & ~(uintptr_t)3);
                    uint64_t const bit0 = 8ull * ((uintptr_t)sbase & 3);
                    // chunks from the stream's end to its start; inside a chunk the higher lanes' symbols come first: a lane's 16 symbols start at
                    // (bits of all later chunks) + (bits of the chunk's higher lanes) = run + chunk total - inclusive prefix
                    const uint8_t* const p = lits + segStart[q];
                    uint32_t const segLen = segLenA[q];
                    uint32_t const nChunks = (segLen + 1023u) >> 10;           // wave-uniform (stream geometry)
                    uint32_t run = 0;
                    for (uint32_t ch = nChunks; ch-- > 0; ) {
                        uint32_t const c0 = (ch << 10) + 16u * (uint32_t)lane;
                        uint32_t const c = c0 < segLen ? (segLen - c0 < 16u ? segLen - c0 : 16u) : 0u;
                        uint32_t cd[16]; uint32_t myBits = 0;
                        if (c) {
                            uint4 va; __builtin_memcpy(&va, p + c0, 16);
                            uint32_t const w[4] = { va.x, va.y, va.z, va.w };
#pragma unroll
                            for (uint32_t b = 0; b < 16; b++) { cd[b] = b < c ? sh->code[(w[b >> 2] >> (8 * (b & 3))) & 0xFF] : 0u; myBits += cd[b] & 0xFF; }
                        }
                        uint32_t const incl = wave_incl_scan(myBits);
                        uint32_t const chunkBits = __shfl(incl, 63);
                        if (c) {
                            RunPacker pk; pk.init(w32, bit0 + (uint64_t)(run + chunkBits - incl));
#pragma unroll
                            for (int b = 15; b >= 1; b -= 2) {                  // two symbols (<= 22 bits) per packer step, last -> first
                                uint32_t const c1 = cd[b], c0v = cd[b - 1];
                                uint32_t const n1 = c1 & 0xFF;
                                pk.add((c1 >> 8) | ((c0v >> 8) << n1), n1 + (c0v & 0xFF));
                            }
                            if (ch == 0 && lane == 0) pk.add(1, 1);             // the stream's FIRST symbols come last: the end mark follows them
                            pk.finish();
                        }
                        run += chunkBits;
                    }
                    if (segLen == 0 && lane == 0) { RunPacker e; e.init(w32, bit0); e.add(1, 1); e.finish(); }      // an empty stream is its end mark
                }
            } else {
#pragma unroll
                for (uint32_t q = 0; q < SPW; q++) {
                    uint32_t const sI = (uint32_t)wv + q * NW;
                    if (sI >= nStreams) continue;
                    uint8_t* const sbase = litDst + sh->streamOff[sI];
                    uint32_t* const w32 = (uint32_t*)((uintptr_t)sbase & ~(uintptr_t)3);
                    uint64_t const bit0 = 8ull * ((uintptr_t)sbase & 3);
#ifndef ZHIP_HUF_PACK4
#define ZHIP_HUF_PACK4 1
#endif
#if ZHIP_HUF_PACK4
                    RunPacker4 pk;
#else
                    RunPacker pk;
#endif
                    pk.init(w32, bit0 + (uint64_t)(total[q] - incl[q]));
                    const uint8_t* p = lits + segStart[q] + runStart[q];
                    // the run is consumed from its end, 32 bytes per round; the first round takes the odd part
                    uint32_t i = runLen[q];
                    while (i) {
                        uint32_t const c = (i & 31) ? (i & 31) : 32;
                        i -= c;
                        uint4 va, vb; __builtin_memcpy(&va, p + i, 16); __builtin_memcpy(&vb, p + i + 16, 16);
                        uint32_t const w[8] = { va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w };
                        for (int b = 31; b >= 1; b -= 2) {                  // two symbols (<= 22 bits) per packer step
                            uint32_t const c1 = (uint32_t)b < c ? sh->code[(w[b >> 2] >> (8 * (b & 3))) & 0xFF] : 0;
                            uint32_t const c0 = (uint32_t)(b - 1) < c ? sh->code[(w[(b - 1) >> 2] >> (8 * ((b - 1) & 3))) & 0xFF] : 0;
                            uint32_t const n1 = c1 & 0xFF;
                            pk.add((c1 >> 8) | ((c0 >> 8) << n1), n1 + (c0 & 0xFF));
                        }
                    }
                    if (lane == 0) pk.add(1, 1);              // lane 0 holds the FIRST symbols = the end of the stream: end mark
                    pk.finish();
                }
            }
            __syncthreads();
        }
    }
    if (litMode != 2) {
        // raw (zstd_compress_literals.c:39) or RLE (:81) literals
        uint32_t const fl = 1 + (litSize > 31) + (litSize > 4095);
        if (t == 0) {
            uint32_t const ty = (litMode == 1) ? 1u : 0u;
            if (fl == 1) litDst[0] = (uint8_t)(ty + (litSize << 3));
            else if (fl == 2) { uint32_t const v = ty + (1 << 2) + (litSize << 4); litDst[0] = (uint8_t)v; litDst[1] = (uint8_t)(v >> 8); }
            else { uint32_t const v = ty + (3 << 2) + (litSize << 4); litDst[0] = (uint8_t)v; litDst[1] = (uint8_t)(v >> 8); litDst[2] = (uint8_t)(v >> 16); litDst[3] = (uint8_t)(v >> 24); }
            if (litMode == 1) litDst[fl] = lits[0];
            sh->litSectionSize = (litMode == 1) ? fl + 1 : fl + litSize;
        }
        if (litMode == 0) for (uint32_t i = (uint32_t)t; i < litSize; i += NT) litDst[fl + i] = lits[i];
        __syncthreads();
    }
    ZPROF(3);
    uint32_t const litSection = sh->litSectionSize;

    // ---------------- sequences section (zstd_compress.c:2934-2997)
    uint8_t* const seqDst = body + litSection;
    uint32_t nbHdr = (nbSeq < 128) ? 1 : (nbSeq < 0x7F00 ? 2 : 3);
    if (t == 0) {
        if (nbSeq < 128) seqDst[0] = (uint8_t)nbSeq;
        else if (nbSeq < 0x7F00) { seqDst[0] = (uint8_t)((nbSeq >> 8) + 0x80); seqDst[1] = (uint8_t)nbSeq; }
        else { seqDst[0] = 0xFF; seqDst[1] = (uint8_t)(nbSeq - 0x7F00); seqDst[2] = (uint8_t)((nbSeq - 0x7F00) >> 8); }
    }
    uint32_t seqSection = nbHdr;
    bool rawBlock = false;
    if (nbSeq > 0) {
        bool const tblFail = (sh->encType[0] == 9) | (sh->encType[1] == 9) | (sh->encType[2] == 9);
        if (tblFail) rawBlock = true;          // cannot happen for valid histograms; keep the frame valid regardless
        if (!rawBlock) {
            // per-sequence bit counts -> positions.  Stream order (LSB first): sequence nbSeq-1 first, then nbSeq-2 ...;
            // inside a sequence: [OF state][ML state][LL state] (not for the first-coded one) [LL extra][ML extra][OF extra].
            // Round 4: COALESCED.  Thread t takes the sequences t, t + NT, ... (one line per wavefront and array instead of 64); the old form gave
            // every thread a run of consecutive sequences, i.e. 64 lines per load instruction, and was 45 % of the stage on dense-match data.
            uint32_t const nbSeqU = ZHIP_UNIFORM(nbSeq);                 // the tile loop below holds barriers: its trip count is scalar
#ifndef ZHIP_SEQ_BITS_FROM_TABLES
#define ZHIP_SEQ_BITS_FROM_TABLES 1
#endif
            // Round 6: the stream's size needs no pass over the sequences — the extra bits are sum(count x bits) per field, the state bits were
            // summed by the chains as they wrote their records (entries 0 .. nbSeq-2: the first-coded sequence has none)
            constexpr bool bitsFromTables = ZHIP_SEQ_BITS_FROM_TABLES != 0;            // (both forms; the one-wavefront form's three chains run side by side and sum together)
            uint32_t myBits = 0;
            if constexpr (!bitsFromTables)
            for (uint32_t i = (uint32_t)t; i < nbSeqU; i += NT) {
                ZhipSeq const sq1 = seqs[i];
                uint32_t const nb1 = (uint32_t)(bLL[i] >> 12) + (bOF[i] >> 12) + (bML[i] >> 12);
                uint32_t ll, mlb, ob; seq_unpack(sq1, pm, i, ll, mlb, ob);
                myBits += TB.llBits[ll_code(TB, ll)] + TB.mlBits[ml_code(TB, mlb)] + hb32(ob);
                if (i + 1 < nbSeqU) myBits += nb1;
            }
            uint32_t totalSeqBits;
            if constexpr (bitsFromTables) totalSeqBits = sh->extraBits[0] + sh->extraBits[1] + sh->extraBits[2] + sh->chainBits[0] + sh->chainBits[1] + sh->chainBits[2];
            else (void)block_excl_scan<NT, SH>(sh, myBits, &totalSeqBits);
            uint32_t const tailBits = sh->ct[2].tableLog + sh->ct[1].tableLog + sh->ct[0].tableLog;
            uint32_t const streamBits = totalSeqBits + tailBits;                       // + 1 end mark
            uint32_t const streamBytes = (streamBits >> 3) + 1;
            uint32_t const tblBytes = sh->ncountSize[0] + sh->ncountSize[1] + sh->ncountSize[2];
            uint8_t* const bs = seqDst + nbHdr + 1 + tblBytes;
            // a block that cannot beat the uncompressed form is emitted raw (:3026): known before a bit is packed, so the
            // bitstream is never written — and never runs past the block's output room
            if (litSection + nbHdr + 1 + tblBytes + streamBytes >= n - minGainBlock) rawBlock = true;
            else {
#ifndef ZHIP_SEQ_STAGE
#define ZHIP_SEQ_STAGE 1
#endif
#if ZHIP_SEQ_STAGE
            // Round 6: every byte of the bitstream is stored once.  Tiles of NT consecutive sequences, from the top of the block down (the highest sequence comes
            // first in the stream); a tile's bits are OR-ed into an LDS image of its stretch of the stream and leave with whole-word stores; the tile's last partial
            // word stays behind as the next image's first (round 5 zeroed the stream in global memory and joined the tiles with atomicOr).  The stream's first word
            // (shared with the table headers) and its last go out byte by byte.  The next tile's four loads per thread are in flight while this one is packed.
            if (t == 0) {
                uint8_t* op = seqDst + nbHdr;
                *op++ = (uint8_t)((sh->encType[0] << 6) + (sh->encType[1] << 4) + (sh->encType[2] << 2));
                for (int k = 0; k < 3; k++) for (uint32_t i = 0; i < sh->ncountSize[k]; i++) *op++ = sh->ncount[k][i];
            }
            {   uint32_t* const w32 = (uint32_t*)((uintptr_t)bs & ~(uintptr_t)3);
                uint8_t* const w8 = (uint8_t*)w32;
                uint32_t const firstB = (uint32_t)((uintptr_t)bs & 3), endB = firstB + streamBytes;      // the stream's bytes are [firstB, endB) counted from w32
                uint32_t* const tile = &sh->hist[0][0];                     // the literal histograms (dead by now): 1 536 words with sampleHist[] behind them
#ifndef ZHIP_SEQ_SPT
#define ZHIP_SEQ_SPT 2
#endif
                // a tile is SPT x NT sequences: thread t takes sequence t of each of the tile's SPT parts (coalesced), one scan (16 bits per part) places all of them,
                // and the barriers of a tile are shared by SPT sequences per thread (SPT 2: at most 31 + 512 x 89 bits = 1 425 words of image)
                constexpr uint32_t SPT = NT == 64 ? 1u : (uint32_t)ZHIP_SEQ_SPT, TS = SPT * NT;     // the one-wavefront form's image is its single histogram: 31 + 64 x 89 bits
                uint32_t const nTiles = (nbSeqU + TS - 1) / TS;
                uint32_t carry = 8u * firstB, wordBase = 0, cw = 0;         // bits of image word 0 in front of the tile's; image word 0 = w32[wordBase]; its value
                ZhipSeq sqN[SPT]; uint32_t xN[SPT], yN[SPT], zN[SPT];
#pragma unroll
                for (uint32_t h = 0; h < SPT; h++) {
                    uint32_t const i = (nTiles - 1) * TS + h * NT + (uint32_t)t;
                    sqN[h] = ZhipSeq{}; xN[h] = yN[h] = zN[h] = 0;
                    if (i < nbSeqU) { sqN[h] = seqs[i]; xN[h] = bOF[i]; yN[h] = bML[i]; zN[h] = bLL[i]; }
                }
#ifdef ZHIP_PROBE_NOSEQPACK        /* timing probe, bytes wrong */
                for (uint32_t k = 0; k-- > 0; ) {
#else
                for (uint32_t k = nTiles; k-- > 0; ) {
#endif
                    uint32_t x[SPT], y[SPT], z[SPT], ll[SPT], mlb[SPT], ob[SPT], lb[SPT], mb[SPT], ofc[SPT], nb[SPT];
                    ZhipSeq sq1[SPT];
                    uint32_t packed = 0;
#pragma unroll
                    for (uint32_t h = 0; h < SPT; h++) {
                        uint32_t const i = k * TS + h * NT + (uint32_t)t;
                        sq1[h] = sqN[h]; x[h] = xN[h]; y[h] = yN[h]; z[h] = zN[h];
                        if (k > 0) { uint32_t const j = i - TS; sqN[h] = seqs[j]; xN[h] = bOF[j]; yN[h] = bML[j]; zN[h] = bLL[j]; }      // every lower tile is full
                        ll[h] = 0; mlb[h] = 0; ob[h] = 1; lb[h] = 0; mb[h] = 0; ofc[h] = 0; nb[h] = 0;
                        if (i < nbSeqU) {
                            seq_unpack(sq1[h], pm, i, ll[h], mlb[h], ob[h]);
                            lb[h] = TB.llBits[ll_code(TB, ll[h])]; mb[h] = TB.mlBits[ml_code(TB, mlb[h])]; ofc[h] = hb32(ob[h]);
                            if (i + 1 >= nbSeqU) { x[h] = 0; y[h] = 0; z[h] = 0; }      // the first-coded sequence carries no state bits
                            nb[h] = (x[h] >> 12) + (y[h] >> 12) + (z[h] >> 12) + lb[h] + mb[h] + ofc[h];
                        }
                        packed |= nb[h] << (16 * h);
                    }
                    uint32_t partTot;
                    uint32_t const lowerP = block_excl_scan<NT, SH>(sh, packed, &partTot);     // per part: bits of its sequences BELOW mine (the barriers also end the last flush)
                    uint32_t tileBits = 0;
#pragma unroll
                    for (uint32_t h = 0; h < SPT; h++) tileBits += (partTot >> (16 * h)) & 0xFFFFu;
                    uint32_t const tot = carry + tileBits, full = tot >> 5, words = (tot + 31) >> 5;
                    for (uint32_t w = (uint32_t)t; w < words; w += NT) tile[w] = w ? 0u : cw;
                    __syncthreads();
                    uint32_t above = 0;                                     // bits of the tile's parts above part h (they come first in the stream)
#pragma unroll
                    for (uint32_t hh = SPT; hh-- > 0; ) {
                        uint32_t const pT = (partTot >> (16 * hh)) & 0xFFFFu, lower = (lowerP >> (16 * hh)) & 0xFFFFu;
                        if (nb[hh]) {
                            // my bits start behind those of the higher sequences: [OF state][ML state][LL state][LL extra][ML extra][OF extra]
                            uint32_t pos = carry + above + (pT - lower - nb[hh]);
                            uint32_t* wp = tile + (pos >> 5); uint64_t acc = 0; uint32_t have = pos & 31;
                            auto put = [&](uint32_t v, uint32_t n) {
                                acc |= (uint64_t)v << have; have += n;
                                if (have >= 32) { __hip_atomic_fetch_or(wp, (uint32_t)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); wp++; acc >>= 32; have -= 32; }
                            };
                            put(x[hh] & 0xFFF, x[hh] >> 12); put(y[hh] & 0xFFF, y[hh] >> 12); put(z[hh] & 0xFFF, z[hh] >> 12);
                            put(ll[hh] & ((1u << lb[hh]) - 1), lb[hh]); put(mlb[hh] & ((1u << mb[hh]) - 1), mb[hh]);
                            put(ob[hh] & (uint32_t)((1ull << ofc[hh]) - 1), ofc[hh]);
                            if (have) __hip_atomic_fetch_or(wp, (uint32_t)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        above += pT;
                    }
                    __syncthreads();
                    for (uint32_t w = (uint32_t)t; w < full; w += NT) {
                        uint32_t const v = tile[w], g = wordBase + w, b0 = 4u * g;
                        if (b0 >= firstB && b0 + 4u <= endB) w32[g] = v;
                        else for (uint32_t q = 0; q < 4; q++) if (b0 + q >= firstB && b0 + q < endB) w8[b0 + q] = (uint8_t)(v >> (8 * q));
                    }
                    cw = (tot & 31) ? tile[full] : 0u;
                    carry = tot & 31; wordBase += full;
                }
                if (t == 0) {     // final states ML, OF, LL then the end mark (zstd_compress_sequences.c:371-381): at most 27 bits behind the carried ones
                    uint64_t acc = cw; uint32_t have = carry;
                    acc |= (uint64_t)(sh->finalState[2] & ((1u << sh->ct[2].tableLog) - 1)) << have; have += sh->ct[2].tableLog;
                    acc |= (uint64_t)(sh->finalState[1] & ((1u << sh->ct[1].tableLog) - 1)) << have; have += sh->ct[1].tableLog;
                    acc |= (uint64_t)(sh->finalState[0] & ((1u << sh->ct[0].tableLog) - 1)) << have; have += sh->ct[0].tableLog;
                    acc |= 1ull << have;
                    uint32_t const b0 = 4u * wordBase;
                    for (uint32_t q = 0; q < 8; q++) if (b0 + q >= firstB && b0 + q < endB) w8[b0 + q] = (uint8_t)(acc >> (8 * q));
                }
            }
#else
            // zero the bytes of the bitstream; lane 0 writes the table headers in front of it
            zero_bytes<NT>(bs, streamBytes);
            if (t == 0) {
                uint8_t* op = seqDst + nbHdr;
                *op++ = (uint8_t)((sh->encType[0] << 6) + (sh->encType[1] << 4) + (sh->encType[2] << 2));
                for (int k = 0; k < 3; k++) for (uint32_t i = 0; i < sh->ncountSize[k]; i++) *op++ = sh->ncount[k][i];
            }
            __syncthreads();
            {   uint32_t* const w32 = (uint32_t*)((uintptr_t)bs & ~(uintptr_t)3);
                uint64_t const bit0 = 8ull * ((uintptr_t)bs & 3);
                // Tiles of NT consecutive sequences, from the top of the block down (the highest sequence comes first in the stream).  Inside a tile
                // sequence i starts where the higher ones of the tile end: a prefix sum over the tile; its bits are OR-ed into an LDS image of the
                // tile's stretch of the stream (aligned like the stream's words), which then goes out with whole-word stores — the first and the last
                // word of a tile are shared with its neighbours and go through atomicOr.  The image lives in the literal histograms (dead by now).
                uint32_t* const tile = &sh->hist[0][0];
                uint32_t const nTiles = (nbSeqU + NT - 1) / NT;
                uint32_t run = 0;                                           // bits placed so far (those of all higher tiles)
                for (uint32_t k = nTiles; k-- > 0; ) {
                    uint32_t const i = k * NT + (uint32_t)t;
                    bool const on = i < nbSeqU;
                    uint32_t x = 0, y = 0, z = 0, ll = 0, mlb = 0, ob = 1, lb = 0, mb = 0, ofc = 0, nb = 0;
                    if (on) {
                        ZhipSeq const sq1 = seqs[i]; x = bOF[i]; y = bML[i]; z = bLL[i];
                        seq_unpack(sq1, pm, i, ll, mlb, ob);
                        lb = TB.llBits[ll_code(TB, ll)]; mb = TB.mlBits[ml_code(TB, mlb)]; ofc = hb32(ob);
                        if (i + 1 >= nbSeqU) { x = 0; y = 0; z = 0; }      // the first-coded sequence carries no state bits
                        nb = (x >> 12) + (y >> 12) + (z >> 12) + lb + mb + ofc;
                    }
                    uint32_t tileBits;
                    uint32_t const lower = block_excl_scan<NT, SH>(sh, nb, &tileBits);        // bits of the tile's sequences BELOW mine
                    uint32_t const sh0 = (uint32_t)((bit0 + run) & 31);
                    uint32_t const words = (sh0 + tileBits + 31) >> 5;
                    for (uint32_t w = (uint32_t)t; w < words; w += NT) tile[w] = 0;
                    __syncthreads();
                    if (nb) {
                        // my bits start behind those of the tile's higher sequences: [OF state][ML state][LL state][LL extra][ML extra][OF extra]
                        uint32_t pos = sh0 + (tileBits - lower - nb);
                        uint32_t* wp = tile + (pos >> 5); uint64_t acc = 0; uint32_t have = pos & 31;
                        auto put = [&](uint32_t v, uint32_t n) {
                            acc |= (uint64_t)v << have; have += n;
                            if (have >= 32) { __hip_atomic_fetch_or(wp, (uint32_t)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); wp++; acc >>= 32; have -= 32; }
                        };
                        put(x & 0xFFF, x >> 12); put(y & 0xFFF, y >> 12); put(z & 0xFFF, z >> 12);
                        put(ll & ((1u << lb) - 1), lb); put(mlb & ((1u << mb) - 1), mb);
                        put(ob & (uint32_t)((1ull << ofc) - 1), ofc);
                        if (have) __hip_atomic_fetch_or(wp, (uint32_t)acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    __syncthreads();
                    {   uint32_t* const g = w32 + ((bit0 + run) >> 5);
                        for (uint32_t w = (uint32_t)t; w < words; w += NT) {
                            uint32_t const v = tile[w];
                            if (v) { if (w == 0 || w + 1 == words) atomicOr(g + w, v); else g[w] = v; }
                        }
                    }
                    run += tileBits;
                    __syncthreads();                                        // the image is free again
                }
                if (t == 0) {     // final states ML, OF, LL then the end mark (zstd_compress_sequences.c:371-381)
                    RunPacker tl; tl.init(w32, bit0 + totalSeqBits);
                    tl.add(sh->finalState[2] & ((1u << sh->ct[2].tableLog) - 1), sh->ct[2].tableLog);
                    tl.add(sh->finalState[1] & ((1u << sh->ct[1].tableLog) - 1), sh->ct[1].tableLog);
                    tl.add(sh->finalState[0] & ((1u << sh->ct[0].tableLog) - 1), sh->ct[0].tableLog);
                    tl.add(1, 1);
                    tl.finish();
                }
            }
#endif
            seqSection = nbHdr + 1 + tblBytes + streamBytes;
            // zstd_compress.c:2987: last FSE header + bitstream < 4 bytes -> emit the block uncompressed
            uint32_t lastCount = 0;
            for (int k = 0; k < 3; k++) if (sh->encType[k] == 2) lastCount = sh->ncountSize[k];
            if (lastCount && lastCount + streamBytes < 4) rawBlock = true;
            __syncthreads();
            }
            ZPROF(6);
        }
    }

    uint32_t const cSize = litSection + seqSection;
#ifdef ZHIP_ENT_DEBUG
    if (t == 0) printf("ENT n=%u nbSeq=%u litSize=%u litMode=%u litSection=%u seqSection=%u enc=%u,%u,%u hufHdr=%u rawBlock=%d\n", n, nbSeq, litSize, sh->litMode, litSection, seqSection, sh->encType[0], sh->encType[1], sh->encType[2], sh->hufHdrSize, (int)rawBlock);
#endif
    if (cSize >= n - minGainBlock) rawBlock = true;         // zstd_compress.c:3026
    ZPROF(7);
    ZPROF_FLUSH(16);
    return rawBlock ? 0u : cSize;
}

// ================================================================== the unit encoder: one frame holding one block
template <uint32_t NT, typename SH>
__device__ inline void entropy_unit(const uint8_t* __restrict__ src, const ZhipUnit& u, const ZhipSeq* __restrict__ seqs,
                                    const ZhipParse& pm, const uint8_t* __restrict__ lits, uint16_t* __restrict__ stBits, uint32_t seqCap,
                                    uint8_t* __restrict__ out, uint32_t* outSize, SH* sh,
                                    const ZhipDictEntropy* __restrict__ de /* dictionary entropy state or nullptr */, uint32_t dictID,
                                    bool withChecksum, uint32_t checksum /* low 32 bits of XXH64(content), zstd_compress.c:5297 */)
{
    int const t = (int)threadIdx.x;
    uint32_t const n = u.srcLen;
    uint32_t const fh = frame_header_size(n, dictID);
    uint8_t* const body = out + fh + 3;                     // block content starts after frame + block header
    uint32_t cSize = 0;
    // trivial units (empty, or too small to attempt compression: zstd_compress.c:3216, :5270) are emitted uncompressed
    if (n >= 7) cSize = entropy_block<NT, SH>(src, n, u, seqs, pm, lits, stBits, seqCap, body, sh, de);
    bool const rawBlock = cSize == 0;
    if (rawBlock) {
        __syncthreads();                                        // thread 0's section-header bytes land before the copy overwrites them
        for (uint32_t i = (uint32_t)t; i < n; i += NT) body[i] = src[i];
    }
    // block + frame headers (zstd_compress.c:4582-4590)
    if (t == 0) {
        write_frame_header(out, n, dictID, withChecksum);
        uint32_t const bh = rawBlock ? (1u + (0u << 1) + (n << 3)) : (1u + (2u << 1) + (cSize << 3));
        out[fh] = (uint8_t)bh; out[fh + 1] = (uint8_t)(bh >> 8); out[fh + 2] = (uint8_t)(bh >> 16);
        uint32_t end = fh + 3 + (rawBlock ? n : cSize);
        if (withChecksum) { for (int b = 0; b < 4; b++) out[end + b] = (uint8_t)(checksum >> (8 * b)); end += 4; }   // zstd_compress.c:5297-5303
        *outSize = end;
    }
}

}  // namespace zhip
