// zhip_decode_big.h — ONE large frame decoded by the whole GPU (the other side of the single-frame / job-pool frames of zhip_frame.h).
//
// WHAT it computes: the content of one RFC 8878 frame without a dictionary (its header may or may not state the content size) — what
// ZSTD_decompressFrame (lib/decompress/zstd_decompress.c:951-1064) regenerates — but block-parallel.  k_decode (zhip_decode.h) walks a
// frame's blocks one after the other in one workgroup (0.26 GB/s on a 1 GiB frame); here the blocks are independent work until the very
// last step.  Anything this path does not like (an error of any kind, a frame shape outside its limits) makes the caller fall back to
// k_decode, which reports the reference's own error codes: this path only has to be exact on valid frames and memory-safe on all.
//
// HOW.  What chains the blocks of a frame together in the reference, and what replaces each chain here:
//   * where a block starts (3-byte headers, zstd_decompress.c:1000-1040)            -> k_bf_walk: one lane follows the headers (3 bytes per
//     128 KB) — or the host does, when the frame lies in host memory (bf_walk_core is the same function on both sides); everything else about a block (literals header, number of sequences, table modes) is read block-parallel by k_bf_prep;
//   * treeless literals / repeat-mode FSE tables (zstd_decompress_block.c:134, :625-660: "the previous block's table")
//                                                                                   -> k_bf_deps: a wave-wide scan names, per block and
//     table, the block whose description DEFINES it; k_bf_entropy rebuilds the table from that block's bytes (a description is a few
//     dozen bytes; re-reading it costs nothing against decoding 128 KB);
//   * the repeat-offset history (ZSTD_decodeSequence :1277-1300)                    -> symbolic: k_bf_entropy starts every block from three
//     MARKERS instead of values; the history arithmetic (swap, rotate, "rep0 - 1") runs on them unchanged — a marker minus k is still a
//     marker — so each record's offset is either a number or "incoming history entry i, minus k", and so is the block's outgoing
//     history: a map closed under composition.  k_bf_scan composes the maps in block order (a few values per block) and hands
//     every block its real incoming history;
//   * where a block's content goes (the sizes are only known after its sequences are decoded) -> the same scan: a prefix sum;
//   * the bytes themselves (ZSTD_execSequence :1001-1095: a match copies what earlier sequences produced, across block borders)
//                                                                                   -> pointer jumping.  k_bf_build writes every LITERAL to its
//     final place and, for every byte a match produces, the position it copies from (map[i] = i - offset; map[i] = i for literals).
//     k_bf_jump replaces map[i] by map[map[i]] (and up to three hops more for entries still on their way) until nothing changes —
//     O(log(longest copy chain)) passes of gathers over the whole frame, every byte of every block at once — and k_bf_copy reads each byte from the literal its chain ends in.  Memory traffic (this
//     machine has 8 TB/s of it) instead of a dependency chain through 8 192 blocks (which nothing hides).
// Algorithmic bytes per frame: compressed in + content out; the map costs 4 bytes per content byte per round on top.
#pragma once
#include "zhip_decode.h"

#define ZHIP_BF_NONE     0xFFFFFFFFu
#define ZHIP_BF_THREADS  128          /* k_bf_entropy: the two wavefronts of decode_frame */
#define ZHIP_BF_SYM      0x80000000u  /* offset values with this bit are symbolic: bits 30:29 = incoming history entry (3 = invalid), low 29 bits = 2^28 - decrements */
#define ZHIP_BF_SYM_ZERO 0x10000000u

struct ZhipBfBlock {              // one per block of the frame
    uint32_t srcOff, csize;       // the block's content (behind its 3-byte header) in the frame; 1 byte for an RLE block
    uint32_t type, rsize;         // 0 raw, 1 RLE, 2 compressed; regenerated size (header value, or computed by k_bf_entropy)
    uint32_t nbSeq, litSize, litType, modes;      // compressed blocks: number of sequences, literals (size, 0 raw / 1 RLE / 2 Huffman / 3 treeless), table modes byte
    uint32_t hufDef, fseDef[3];   // the blocks whose descriptions define the Huffman / LL / OF / ML tables this block uses (itself when it carries one)
    uint32_t repOut[3], repIn[3]; // the offset history it leaves (symbolic) and the one it starts from (resolved by k_bf_scan)
    uint32_t outOff, status;      // where its content starts in the frame; first error
    uint64_t recOff, litOff;      // its records (nbSeq + 1) and its decoded literals in the arenas
};
struct ZhipBfInfo {               // one per frame
    uint32_t nBlocks, status, endPos, checksum;
    uint64_t totalRecs, totalLit, totalOut;
    uint32_t changed, pad;
};

namespace zhip {

// host: the frame header fields this path needs (zstd_decompress.c:438-545); ok = a frame it may take: no dictionary, content size stated
struct BfHeader { uint32_t hdrSize, blockMax, hasChecksum; uint64_t fcs; bool known /* the header states the content size */; bool ok; };
inline BfHeader bf_parse_header(const uint8_t* p, size_t n)
{
    BfHeader h; h.hdrSize = 0; h.blockMax = ZHIP_UNIT_MAX; h.hasChecksum = 0; h.fcs = 0; h.known = false; h.ok = false;
    if (n < 8 || p[0] != 0x28 || p[1] != 0xB5 || p[2] != 0x2F || p[3] != 0xFD) return h;
    uint32_t const fhd = p[4], didCode = fhd & 3, fcsCode = fhd >> 6, single = (fhd >> 5) & 1;
    if ((fhd & 8) || didCode) return h;
    uint32_t const fcsB = fcsCode == 0 ? (single ? 1u : 0u) : (1u << fcsCode);
    h.known = fcsB != 0;                                        // a frame of a streaming compressor may not state its size: then the blocks' sizes say it (k_bf_scan)
    h.hdrSize = 5 + (single ? 0u : 1u) + fcsB;
    if (n < h.hdrSize + 3) return h;
    uint32_t pos = 5; uint64_t window = 0;
    if (!single) { uint32_t const wl = (p[pos] >> 3) + 10; if (wl > 31) return h; window = 1ull << wl; window += (window >> 3) * (p[pos] & 7); pos++; }
    uint64_t v = 0; for (uint32_t i = 0; i < fcsB; i++) v |= (uint64_t)p[pos + i] << (8 * i);
    if (fcsCode == 1) v += 256;
    h.fcs = v; if (single) window = v;
    h.blockMax = window < ZHIP_UNIT_MAX ? (uint32_t)window : ZHIP_UNIT_MAX;
    h.hasChecksum = (fhd >> 2) & 1;
    h.ok = true;
    return h;
}

// ------------------------------------------------------------------ the block headers (zstd_decompress.c:1000-1040): the one serial chain of the frame.
// On the device one lane follows it (k_bf_walk: 3 bytes per block, an HBM round trip each — 7 ms for the 11 776 blocks of a 1 GiB frame); when the
// caller's frame lies in HOST memory (zhip_decompress) the host walks it in microseconds and uploads the table (bigframe_decode).
__host__ __device__ inline void bf_walk_core(const uint8_t* __restrict__ src, uint32_t srcLen, uint32_t hdrSize, uint32_t blockMax, uint32_t hasChecksum,
                                             ZhipBfBlock* __restrict__ blocks, uint32_t capBlocks, ZhipBfInfo* __restrict__ info)
{
    uint32_t ip = hdrSize, nb = 0, status = 0;
    for (;;) {
        if (srcLen - ip < 3) { status = ZHIP_DE_SRC_WRONG; break; }
        uint32_t const bh = src[ip] | (src[ip + 1] << 8) | (src[ip + 2] << 16);
        uint32_t const type = (bh >> 1) & 3, bsize = bh >> 3, csize = type == 1 ? 1u : bsize;
        ip += 3;
        if (type == 3) { status = ZHIP_DE_CORRUPT; break; }
        if (csize > srcLen - ip) { status = ZHIP_DE_SRC_WRONG; break; }
        if (type == 2 && csize > blockMax) { status = ZHIP_DE_SRC_WRONG; break; }
        if (nb == capBlocks) { status = ZHIP_DE_UNSUPPORTED; break; }
        ZhipBfBlock b; __builtin_memset(&b, 0, sizeof(b));
        b.srcOff = ip; b.csize = csize; b.type = type; b.rsize = type == 2 ? 0u : bsize;
        b.hufDef = ZHIP_BF_NONE; b.fseDef[0] = b.fseDef[1] = b.fseDef[2] = ZHIP_BF_NONE;
        blocks[nb++] = b;
        ip += csize;
        if (bh & 1) break;
    }
    uint32_t ck = 0;
    if (!status && hasChecksum) {
        if (srcLen - ip < 4) status = ZHIP_DE_CHECKSUM;
        else ck = (uint32_t)src[ip] | ((uint32_t)src[ip + 1] << 8) | ((uint32_t)src[ip + 2] << 16) | ((uint32_t)src[ip + 3] << 24);
    }
    info->nBlocks = nb; info->status = status; info->endPos = ip; info->checksum = ck; info->changed = 0;
}

#ifndef ZHIP_DECODE_HOST_ONLY
// ------------------------------------------------------------------ k_bf_walk: the block headers, one lane
__device__ inline void bf_walk(const uint8_t* __restrict__ src, uint32_t srcLen, uint32_t hdrSize, uint32_t blockMax, uint32_t hasChecksum,
                               ZhipBfBlock* __restrict__ blocks, uint32_t capBlocks, ZhipBfInfo* __restrict__ info)
{
    if (threadIdx.x != 0) return;
    bf_walk_core(src, srcLen, hdrSize, blockMax, hasChecksum, blocks, capBlocks, info);
}

// ------------------------------------------------------------------ k_bf_prep: one thread per block
__device__ inline void bf_prep(const uint8_t* __restrict__ src, uint32_t blockMax, ZhipBfBlock* __restrict__ blocks, uint32_t bi)
{
    ZhipBfBlock b = blocks[bi];
    if (b.type != 2) return;
    const uint8_t* const blk = src + b.srcOff;
    LitHeader const h = dec_lit_header(blk, b.csize, blockMax);
    uint32_t err = h.err;
    if (!err) {
        b.litType = h.type; b.litSize = h.litSize;
        uint32_t const secOff = h.lh + h.cSize;
        const uint8_t* const seq = blk + secOff; uint32_t const size = b.csize - secOff;
        uint32_t pos = 0, nbSeq = 0;
        if (size < 1) err = ZHIP_DE_SRC_WRONG;
        else {
            nbSeq = seq[pos++];
            if (nbSeq > 0x7F) {
                if (nbSeq == 0xFF) { if (pos + 2 > size) err = ZHIP_DE_SRC_WRONG; else { nbSeq = (seq[pos] | (seq[pos + 1] << 8)) + 0x7F00; pos += 2; } }
                else { if (pos >= size) err = ZHIP_DE_SRC_WRONG; else nbSeq = ((nbSeq - 0x80) << 8) + seq[pos++]; }
            }
        }
        if (!err && nbSeq == 0 && pos != size) err = ZHIP_DE_CORRUPT;
        if (!err && nbSeq) { if (pos + 1 > size) err = ZHIP_DE_SRC_WRONG; else { b.modes = seq[pos]; if (b.modes & 3) err = ZHIP_DE_CORRUPT; } }
        b.nbSeq = err ? 0 : nbSeq;
    }
    b.status = err;
    blocks[bi] = b;
}

// ------------------------------------------------------------------ k_bf_deps: one wavefront; defining blocks of the repeated tables, arena offsets
__device__ inline void bf_deps(ZhipBfBlock* __restrict__ blocks, ZhipBfInfo* __restrict__ info)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const nB = info->nBlocks;
    uint32_t lastHuf = ZHIP_BF_NONE, lastFse[3] = { ZHIP_BF_NONE, ZHIP_BF_NONE, ZHIP_BF_NONE }, status = info->status;
    uint64_t recs = 0, lits = 0;
    for (uint32_t b0 = 0; b0 < nB; b0 += 64) {
        uint32_t const bi = b0 + lane; bool const on = bi < nB;
        ZhipBfBlock b; __builtin_memset(&b, 0, sizeof(b));
        if (on) b = blocks[bi];
        bool const comp = on && b.type == 2 && b.status == 0;
        unsigned long long const bad = __ballot(on && b.status != 0);
        if (bad && !status) status = __builtin_amdgcn_readlane(b.status, first_lane(bad));
        unsigned long long const below = below_mask((int)lane + 1);
        // Huffman: a type-2 literals section defines, a type-3 one uses the latest definition
        {   unsigned long long const def = __ballot(comp && b.litType == 2) ;
            unsigned long long const mine = def & below;
            uint32_t const d = mine ? b0 + 63u - (uint32_t)__clzll((long long)mine) : lastHuf;
            if (comp && b.litType >= 2) { b.hufDef = d; if (d == ZHIP_BF_NONE && !status) status = ZHIP_DE_DICT_CORRUPT; }
            if (def) lastHuf = b0 + 63u - (uint32_t)__clzll((long long)def);
        }
        for (int k = 0; k < 3; k++) {
            uint32_t const mode = (b.modes >> (6 - 2 * k)) & 3;
            bool const seqs = comp && b.nbSeq > 0;
            unsigned long long const def = __ballot(seqs && mode != 3);
            unsigned long long const mine = def & below;
            uint32_t const d = mine ? b0 + 63u - (uint32_t)__clzll((long long)mine) : lastFse[k];
            if (seqs) { b.fseDef[k] = d; if (d == ZHIP_BF_NONE && !status) status = ZHIP_DE_CORRUPT; }
            if (def) lastFse[k] = b0 + 63u - (uint32_t)__clzll((long long)def);
        }
        status = __builtin_amdgcn_readfirstlane(__ballot(status != 0) ? __builtin_amdgcn_readlane(status, first_lane(__ballot(status != 0))) : 0u);
        // arena offsets: nbSeq + 1 records per compressed block; literals of Huffman-coded sections
        uint32_t const nr = comp ? b.nbSeq + 1 : 0, nl = (comp && b.litType >= 2) ? ((b.litSize + 63u) & ~63u) + 64u : 0;
        uint32_t incR = nr, incL = nl;
        for (int sft = 1; sft < 64; sft <<= 1) { uint32_t const a = __shfl_up(incR, (unsigned)sft), c = __shfl_up(incL, (unsigned)sft); if ((int)lane >= sft) { incR += a; incL += c; } }
        b.recOff = recs + (incR - nr); b.litOff = lits + (incL - nl);
        recs += __builtin_amdgcn_readlane(incR, 63); lits += __builtin_amdgcn_readlane(incL, 63);
        if (on) blocks[bi] = b;
    }
    if (lane == 0) { info->status = status; info->totalRecs = recs; info->totalLit = lits; }
}

// ------------------------------------------------------------------ k_bf_entropy: one 128-thread workgroup per block
// descriptors of a sequences section up to table k (lane 0): returns the position behind them, or 0 on error; the description of
// table `want` (0..2, or 3 = none) is left in hdr / rle / mx / lg and, when it is FSE-compressed, its distribution in S->norm[want]
struct BfTab { uint32_t hdr, rle, mx, lg; };
__device__ inline uint32_t bf_walk_tables(DecShared* S, const uint8_t* seq, uint32_t size, uint32_t upTo /* tables 0 .. upTo-1 are walked */, uint32_t want, BfTab* T)
{
    uint32_t pos = 0;
    if (size < 1) return 0;
    uint32_t nbSeq = seq[pos++];
    if (nbSeq > 0x7F) pos += nbSeq == 0xFF ? 2 : 1;
    if (pos + 1 > size) return 0;
    uint32_t const modes = seq[pos++];
    for (uint32_t k = 0; k < upTo; k++) {
        uint32_t const type = (modes >> (6 - 2 * k)) & 3;
        BfTab t; t.hdr = 0; t.rle = 0; t.mx = 0; t.lg = 0;
        if (type == 0) { t.hdr = 3; t.lg = k == 1 ? 5 : 6; }
        else if (type == 1) { if (pos >= size || seq[pos] > dec_max_sym((int)k)) return 0; t.hdr = 2; t.rle = seq[pos++]; }
        else if (type == 2) {
            uint32_t m = dec_max_sym((int)k), tl = 0;
            uint32_t const h = fse_d_read_ncount(S->norm[k], &m, &tl, seq + pos, size - pos);      // tables below `want` are built already: their norm[] is scratch
            if (!h || tl > dec_max_log((int)k)) return 0;
            t.hdr = 1; t.mx = m; t.lg = tl; pos += h;
        } else t.hdr = 0;                                        // repeat: described elsewhere
        if (k == want) *T = t;
    }
    return pos ? pos : 0;
}

__device__ inline void bf_entropy_block(DecShared* S, const uint8_t* __restrict__ src, uint32_t blockMax, ZhipBfBlock* __restrict__ blocks, uint32_t bi,
                                        uint8_t* __restrict__ litArena, ZhipDSeq* __restrict__ recArena, const uint64_t* __restrict__ defTabs)
{
    uint32_t const tid = threadIdx.x, wave = ZHIP_UNIFORM(tid >> 6), lane = tid & 63;       // scalar branches around the barriers (zhip_common.h, ZHIP_UNIFORM)
    ZhipBfBlock const b = blocks[bi];
    if (ZHIP_UNIFORM(b.type) != 2) {                                          // raw / RLE: nothing to decode, the history passes through
        if (tid == 0) { blocks[bi].repOut[0] = ZHIP_BF_SYM | ZHIP_BF_SYM_ZERO; blocks[bi].repOut[1] = ZHIP_BF_SYM | (1u << 29) | ZHIP_BF_SYM_ZERO; blocks[bi].repOut[2] = ZHIP_BF_SYM | (2u << 29) | ZHIP_BF_SYM_ZERO; }
        return;
    }
    const uint8_t* const blk = src + b.srcOff;
    uint8_t* const lit = litArena + b.litOff;
    ZhipDSeq* const recs = recArena + b.recOff;
    if (tid == 0) { S->status = 0; S->hufValid = 0; S->fseValid = 1; S->nbSeq = b.nbSeq; S->endOut = 0; S->endLit = 0; S->dictHufIn = 0; S->dictFseIn = 0;
                    S->rep[0] = ZHIP_BF_SYM | ZHIP_BF_SYM_ZERO; S->rep[1] = ZHIP_BF_SYM | (1u << 29) | ZHIP_BF_SYM_ZERO; S->rep[2] = ZHIP_BF_SYM | (2u << 29) | ZHIP_BF_SYM_ZERO; }
    ZHIP_CONVERGE();
    __syncthreads();
    LitHeader const lh = dec_lit_header(blk, b.csize, blockMax);
    if (wave == 0) {
        // literals (ZSTD_decodeLiteralsBlock :134-345); a treeless section takes its table from the block that described it
        uint32_t err = 0;
        if (lh.type >= 2) {
            const uint8_t* hs = blk + lh.lh; uint32_t hn = lh.cSize;
            if (lh.type == 2) {
                uint32_t const t = dec_huf_table(S, hs, hn);
                if (!t || t >= hn) err = ZHIP_DE_CORRUPT; else { hs += t; hn -= t; }
            } else {
                ZhipBfBlock const d = blocks[b.hufDef];
                const uint8_t* const dblk = src + d.srcOff;
                LitHeader const dh = dec_lit_header(dblk, d.csize, blockMax);
                uint32_t const t = dh.err ? 0 : dec_huf_table(S, dblk + dh.lh, dh.cSize);
                if (!t) err = ZHIP_DE_CORRUPT;
            }
            if (!err) err = dec_huf_streams_par(S, hs, hn, lh.litSize, lh.single != 0, lit);
        }
        if (err && lane == 0) atomicMax(&S->status, err);
    } else {
        // sequences: the three tables (own description, or the defining block's), then every sequence into the block's records
        uint32_t const secOff = lh.lh + lh.cSize;
        const uint8_t* const seq = blk + secOff; uint32_t const size = b.csize - secOff;
        uint32_t err = 0;
        if (b.nbSeq) {
            uint32_t ownEnd = 0;
            for (uint32_t k = 0; k < 3 && !err; k++) {
                uint32_t const mode = (b.modes >> (6 - 2 * k)) & 3;
                BfTab T; T.hdr = 0; T.rle = 0; T.mx = 0; T.lg = 0;
                uint32_t ok = 1;
                if (lane == 0) {
                    if (mode != 3) ok = bf_walk_tables(S, seq, size, k + 1, k, &T);
                    else {
                        ZhipBfBlock const d = blocks[b.fseDef[k]];
                        const uint8_t* const dblk = src + d.srcOff;
                        LitHeader const dh = dec_lit_header(dblk, d.csize, blockMax);
                        uint32_t const dOff = dh.lh + dh.cSize;
                        ok = dh.err ? 0 : bf_walk_tables(S, dblk + dOff, d.csize - dOff, k + 1, k, &T);
                    }
                }
                ok = __builtin_amdgcn_readfirstlane(ok);
                if (!ok) { err = ZHIP_DE_CORRUPT; break; }
                uint32_t const a = __builtin_amdgcn_readfirstlane(T.hdr), r = __builtin_amdgcn_readfirstlane(T.rle);
                uint32_t const m = __builtin_amdgcn_readfirstlane(T.mx), l = __builtin_amdgcn_readfirstlane(T.lg);
                __builtin_amdgcn_wave_barrier();
                if (a == 3) { uint32_t const n = 1u << l; const uint64_t* const from = defTabs + (k == 0 ? 0 : k == 1 ? 64 : 96); for (uint32_t i = lane; i < n; i += 64) dec_tab(S, k)[i] = from[i]; }
                else if (a == 1) fse_d_build_wave(S, k, m, l);
                else if (a == 2) { if (lane == 0) { uint32_t base, bits; dec_base_bits((int)k, r, &base, &bits); dec_tab(S, k)[0] = fse_d_pack(0, bits, 0, base); } }
                else err = ZHIP_DE_CORRUPT;
                if (lane == 0) S->log[k] = l;
                __builtin_amdgcn_wave_barrier();
            }
            if (!err) {                                         // where the bitstream starts: behind this block's own descriptions
                uint32_t e = 0;
                if (lane == 0) { BfTab T; e = bf_walk_tables(S, seq, size, 3, 3, &T); }
                ownEnd = __builtin_amdgcn_readfirstlane(e);
                if (!ownEnd || ownEnd >= size) err = ZHIP_DE_CORRUPT;
            }
            SeqDec D; D.done = 0; D.outPos = 0; D.litPos = 0; D.sLL = D.sOF = D.sML = 0; D.base = seq; D.size = 0; D.Dpos = 0; D.wLoaded = 0;
            D.rep0 = ZHIP_BF_SYM | ZHIP_BF_SYM_ZERO; D.rep1 = ZHIP_BF_SYM | (1u << 29) | ZHIP_BF_SYM_ZERO; D.rep2 = ZHIP_BF_SYM | (2u << 29) | ZHIP_BF_SYM_ZERO;
            if (!err) {
                uint32_t const bsz = size - ownEnd, lastByte = seq[size - 1];
                if (lastByte == 0) err = ZHIP_DE_CORRUPT;
                else {
                    D.base = seq + ownEnd; D.size = bsz; D.wLoaded = ZHIP_DEC_RING_WORDS;
                    uint32_t const need = (((bsz + 3) >> 2) + 4 + 63) & ~63u;
                    seq_ring_fill(S, D, 0, need < ZHIP_DEC_RING_WORDS ? need : ZHIP_DEC_RING_WORDS);
                    __builtin_amdgcn_wave_barrier();
                    const lds_u32* const R = (const lds_u32*)(uintptr_t)S->ring;
                    uint32_t const l0 = S->log[0], l1 = S->log[1], l2 = S->log[2];
                    uint32_t d = 8 - dec_hb(lastByte);
                    D.sLL = seq_field(R, d, l0); d += l0;
                    D.sOF = seq_field(R, d, l1); d += l1;
                    D.sML = seq_field(R, d, l2); d += l2;
                    D.Dpos = d;
                }
            }
            if (err) { if (lane == 0) atomicMax(&S->status, err); }
            else {
                uint32_t const nChunks = (b.nbSeq + ZHIP_DEC_CHUNK - 1) / ZHIP_DEC_CHUNK;
                for (uint32_t c = 0; c < nChunks; c++) {
                    dec_seq_chunk(S, D, recs + (size_t)c * ZHIP_DEC_CHUNK, 0, b.nbSeq, lh.litSize, blockMax, 0, true);
                    __builtin_amdgcn_wave_barrier();           // lane 0 wrote the count
                    if (S->cnt[0] == 0) break;                 // an error: the status is set
                }
            }
        } else if (lane == 0) { ZhipDSeq r; r.outPos = 0; r.litPos = 0; r.off = 0; r.ml = 0; recs[0] = r; }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t st = S->status;
        uint32_t const endOut = b.nbSeq ? S->endOut : 0, endLit = b.nbSeq ? S->endLit : 0;
        uint32_t rsize = 0;
        if (!st) { if (endLit > lh.litSize) st = ZHIP_DE_CORRUPT; else { rsize = endOut + (lh.litSize - endLit); if (rsize > blockMax) st = ZHIP_DE_UNSUPPORTED; } }
        blocks[bi].rsize = rsize; blocks[bi].status = st;
        blocks[bi].repOut[0] = S->rep[0]; blocks[bi].repOut[1] = S->rep[1]; blocks[bi].repOut[2] = S->rep[2];
    }
}

// ------------------------------------------------------------------ k_bf_scan: one wavefront; output offsets + the offset history in block order
__device__ __forceinline__ uint32_t bf_resolve(uint32_t v, const uint32_t R[3])          // 0 = invalid
{
    if (!(v & ZHIP_BF_SYM)) return v;
    uint32_t const idx = (v >> 29) & 3, k = ZHIP_BF_SYM_ZERO - (v & 0x1FFFFFFFu);
    if (idx == 3) return 0;
    uint32_t const r = R[idx];
    return r > k ? r - k : 0;
}
__device__ inline void bf_scan(ZhipBfBlock* __restrict__ blocks, ZhipBfInfo* __restrict__ info, uint32_t dstCap, uint32_t* sh /* 64 * 3 words of LDS */)
{
    uint32_t const lane = (uint32_t)lane_id();
    uint32_t const nB = info->nBlocks;
    uint32_t status = info->status;
    uint32_t R[3] = { 1, 4, 8 };                                // zstd_decompress.c: repStartValue
    uint64_t out = 0;
    for (uint32_t b0 = 0; b0 < nB; b0 += 64) {
        uint32_t const bi = b0 + lane; bool const on = bi < nB;
        uint32_t rs = 0, st = 0, ro[3] = { 0, 0, 0 };
        if (on) { rs = blocks[bi].rsize; st = blocks[bi].status; ro[0] = blocks[bi].repOut[0]; ro[1] = blocks[bi].repOut[1]; ro[2] = blocks[bi].repOut[2]; }
        unsigned long long const bad = __ballot(on && st != 0);
        if (bad && !status) status = __builtin_amdgcn_readlane(st, first_lane(bad));
        uint32_t inc = rs;
        for (int sft = 1; sft < 64; sft <<= 1) { uint32_t const a = __shfl_up(inc, (unsigned)sft); if ((int)lane >= sft) inc += a; }
        uint64_t const mine = out + (inc - rs);
        if (on) { if (mine + rs > dstCap) st = ZHIP_DE_DST_SMALL; blocks[bi].outOff = (uint32_t)mine; }
        unsigned long long const bad2 = __ballot(on && st == ZHIP_DE_DST_SMALL);
        if (bad2 && !status) status = ZHIP_DE_DST_SMALL;
        out += __builtin_amdgcn_readlane(inc, 63);
        (void)sh;
        uint32_t const cnt = nB - b0 < 64 ? nB - b0 : 64;
        // the composition is a serial walk over values that sit in the lanes' registers (block j's outgoing history is read with a
        // readlane: no memory round trip inside the chain); every lane runs it, lane j keeps what block j starts from
        uint32_t myIn[3] = { 0, 0, 0 };
        for (uint32_t j = 0; j < cnt; j++) {
            if (j == lane) { myIn[0] = R[0]; myIn[1] = R[1]; myIn[2] = R[2]; }
            uint32_t const o0 = __builtin_amdgcn_readlane(ro[0], (int)j), o1 = __builtin_amdgcn_readlane(ro[1], (int)j), o2 = __builtin_amdgcn_readlane(ro[2], (int)j);
            uint32_t const n0 = bf_resolve(o0, R), n1 = bf_resolve(o1, R), n2 = bf_resolve(o2, R);
            if ((n0 == 0 || n1 == 0 || n2 == 0) && !status) status = ZHIP_DE_CORRUPT;
            R[0] = n0; R[1] = n1; R[2] = n2;
        }
        if (on) { blocks[bi].repIn[0] = myIn[0]; blocks[bi].repIn[1] = myIn[1]; blocks[bi].repIn[2] = myIn[2]; }
    }
    if (lane == 0) { info->status = status; info->totalOut = out; }
}

// ------------------------------------------------------------------ k_bf_build: literals to their place, the copy map; one 256-thread workgroup per block
__device__ inline void bf_build_block(const uint8_t* __restrict__ src, const ZhipBfBlock* __restrict__ blocks, uint32_t bi, const uint8_t* __restrict__ litArena,
                                      const ZhipDSeq* __restrict__ recArena, uint8_t* __restrict__ out, uint32_t* __restrict__ map, ZhipBfInfo* __restrict__ info)
{
    uint32_t const tid = threadIdx.x, nT = blockDim.x, lane = tid & 63, wave = tid >> 6, nW = nT >> 6;
    ZhipBfBlock const b = blocks[bi];
    uint32_t const o0 = b.outOff;
    if (b.type != 2) {
        for (uint32_t i = tid; i < b.rsize; i += nT) { out[o0 + i] = b.type == 0 ? src[b.srcOff + i] : src[b.srcOff]; map[o0 + i] = o0 + i; }
        return;
    }
    const uint8_t* const blk = src + b.srcOff;
    LitHeader const lh = dec_lit_header(blk, b.csize, 0xFFFFFFFFu);
    const uint8_t* const litp = lh.type == 0 ? blk + lh.lh : litArena + b.litOff;
    bool const litRle = lh.type == 1;
    uint32_t const litByte = litRle ? blk[lh.lh] : 0;
    const ZhipDSeq* const recs = recArena + b.recOff;
    uint32_t const R[3] = { b.repIn[0], b.repIn[1], b.repIn[2] };
    bool bad = false;
    for (uint32_t j0 = wave * 64; j0 < b.nbSeq; j0 += nW * 64) {
        uint32_t const j = j0 + lane; bool const on = j < b.nbSeq;
        ZhipDSeq r; r.outPos = 0; r.litPos = 0; r.off = 0; r.ml = 0; uint32_t nextLit = 0;
        if (on) { uint4 const v = *(const uint4*)(recs + j); r.outPos = v.x; r.litPos = v.y; r.off = v.z; r.ml = v.w; nextLit = recs[j + 1].litPos; }
        uint32_t const ll = on ? nextLit - r.litPos : 0;
        uint32_t const o = o0 + r.outPos, mo = o + ll;                       // literals at o, the match at mo
        uint32_t const off = on ? bf_resolve(r.off, R) : 1;
        if (on && (off == 0 || off > mo || r.outPos + ll + r.ml > b.rsize)) bad = true;
        bool const ok = on && !(off == 0 || off > mo || r.outPos + ll + r.ml > b.rsize);
        // short runs by their own lane, long ones by the whole wavefront
        if (ok && ll <= 64) for (uint32_t t = 0; t < ll; t++) { out[o + t] = litRle ? (uint8_t)litByte : litp[r.litPos + t]; map[o + t] = o + t; }
        if (ok && r.ml <= 64) for (uint32_t t = 0; t < r.ml; t++) map[mo + t] = mo + t - off;
        unsigned long long longs = __ballot(ok && (ll > 64 || r.ml > 64));
        while (longs) {
            int const f = first_lane(longs); longs &= longs - 1;
            uint32_t const fo = __builtin_amdgcn_readlane(o, f), fll = __builtin_amdgcn_readlane(ll, f), fml = __builtin_amdgcn_readlane(r.ml, f);
            uint32_t const flp = __builtin_amdgcn_readlane(r.litPos, f), foff = __builtin_amdgcn_readlane(off, f);
            if (fll > 64) for (uint32_t t = lane; t < fll; t += 64) { out[fo + t] = litRle ? (uint8_t)litByte : litp[flp + t]; map[fo + t] = fo + t; }
            if (fml > 64) for (uint32_t t = lane; t < fml; t += 64) map[fo + fll + t] = fo + fll + t - foff;
        }
    }
    // the block's last literals (zstd_decompress_block.c:1681-1690)
    {   ZhipDSeq const e = recs[b.nbSeq];
        uint32_t const rest = lh.litSize - e.litPos, o = o0 + e.outPos;
        if (e.outPos + rest != b.rsize) bad = true;
        else for (uint32_t t = tid; t < rest; t += nT) { out[o + t] = litRle ? (uint8_t)litByte : litp[e.litPos + t]; map[o + t] = o + t; }
    }
    if (bad) info->status = ZHIP_DE_CORRUPT;
}

// ------------------------------------------------------------------ k_bf_jump / k_bf_copy: every byte of the frame
__device__ inline void bf_jump(uint32_t* __restrict__ map, uint32_t n, uint32_t* __restrict__ changed)
{
    uint32_t const i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (i0 >= n) return;
    uint4 const v = *(const uint4*)(map + i0);                   // the map has 16 bytes of slack behind entry n - 1
    uint32_t const j4[4] = { v.x, v.y, v.z, v.w };
    uint32_t jj4[4];
    for (uint32_t e = 0; e < 4; e++) jj4[e] = (i0 + e < n && j4[e] != i0 + e) ? map[j4[e]] : j4[e];       // four gathers in flight
    // up to three more hops for the entries that are still on their way (a pass then multiplies a chain's reach by up to five instead of two:
    // every pass reads the whole map once, so passes are what costs once most entries have arrived)
    bool act[4];
    for (uint32_t e = 0; e < 4; e++) act[e] = jj4[e] != j4[e];
    for (int hop = 0; hop < 3; hop++) {
        bool moved = false;
        uint32_t t4[4];
        for (uint32_t e = 0; e < 4; e++) t4[e] = act[e] ? map[jj4[e]] : jj4[e];
        for (uint32_t e = 0; e < 4; e++) { act[e] = act[e] && t4[e] != jj4[e]; if (act[e]) { jj4[e] = t4[e]; moved = true; } }
        if (!moved) break;
    }
    bool any = false;
    for (uint32_t e = 0; e < 4; e++) if (jj4[e] != j4[e]) { map[i0 + e] = jj4[e]; any = true; }
    if (any) *changed = 1;
}
__device__ inline void bf_copy(const uint32_t* __restrict__ map, uint8_t* __restrict__ out, uint32_t n)
{
    uint32_t const i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (i0 >= n) return;
    uint4 const v = *(const uint4*)(map + i0);
    uint32_t const j4[4] = { v.x, v.y, v.z, v.w };
    uint8_t b4[4];
    for (uint32_t e = 0; e < 4; e++) b4[e] = (i0 + e < n && j4[e] != i0 + e) ? out[j4[e]] : 0;
    for (uint32_t e = 0; e < 4; e++) if (i0 + e < n && j4[e] != i0 + e) out[i0 + e] = b4[e];
}
#endif  // ZHIP_DECODE_HOST_ONLY

}  // namespace zhip
