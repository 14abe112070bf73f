// zhip_frame.h — one standard frame with as many blocks as the reference emits for the same input
// (ZSTD_compress_frameChunk, lib/compress/zstd_compress.c:4520-4640; strategies ZSTD_fast and ZSTD_dfast, no dictionary).
//
// What chains the blocks of a frame together, and therefore what this kernel carries from one block to the next:
//   * the match finder's hash table and the window (a match may reach back 2^windowLog bytes into earlier blocks),
//   * the repcode history and the literals' Huffman table — both only when the block was emitted compressed
//     (ZSTD_blockState_confirmRepcodesAndEntropyTables, :3312-3320),
//   * `savings`, which moves the block boundary from 128 KB to 92 KB (ZSTD_optimalBlockSize, :4494-4518).
// The chain is strictly serial, so a frame is one 256-thread workgroup: wave 0 runs the ZSTD_fast parser of zhip_parse.h (hashLog <= 14:
// table in LDS, 24-bit entries when the positions fit — two workgroups per CU — else 32-bit; larger tables in HBM) or the ZSTD_dfast
// parser of zhip_parse_dfast.h (two tables of plain 32-bit positions in HBM) on the block, then the four waves run the block
// encoder of zhip_entropy.h with the previous block's Huffman table as the "repeat" candidate.  Independent frames of a
// batch run side by side, one workgroup each — and so do the JOBS of one frame when it is compressed the way the reference's
// job pool compresses it (ZhipJob below).
#pragma once
#include "zhip_parse.h"
#include "zhip_parse_dfast.h"
#include "zhip_entropy.h"

namespace zhip {

#define ZHIP_FRAME_BLOCK_SPLIT   (92u * 1024u)     /* zstd_compress.c:4517 "blind" split of the strategies below lazy2 */
#ifndef ZHIP_FRAME_LDS_HASHLOG
#define ZHIP_FRAME_LDS_HASHLOG   14u               /* tables up to 64 KB live in LDS */
#endif

struct FrameShared {
    ZhipParse meta;            // the parser's result for the block in flight
    uint32_t  flag;
    uint32_t  pad[5];
    alignas(16) unsigned char dfScratch[2 * ZHIP_DF_SCRATCH];   // ZSTD_dfast: the parser's in-batch duplicate detectors
};

// what one block hands to the next, per frame, in HBM
struct ZhipFrameState { ZhipDictEntropy ent; uint32_t predicted; uint32_t pad[3]; };     // predicted: k_lz_predict marked this window (zhip_frame_lazy.h)

// One JOB of a frame compressed the way ZSTD_c_nbWorkers >= 1 compresses it (zstdmt_compress.c:683-790): a section of the input that
// starts from a fresh context which has only loaded the `prefixLen` bytes in front of it (the overlap with the previous job) as a
// raw-content prefix; jobs are independent of each other, so a frame's jobs run side by side, one workgroup each, and their blocks,
// concatenated in job order, are the frame.  `start` is relative to the frame start; inside the kernel a job counts positions from
// the start of its own window (see frame_fast), so that they fit the 24-bit LDS table whatever the frame's size.
#define ZHIP_JOB_FIRST 1u          /* writes the frame header; repcodes start at {1,4,8} */
#define ZHIP_JOB_LAST  2u          /* its final block carries the last-block bit; the frame checksum follows */
#define ZHIP_JOB_CHUNK (512u * 1024u)   /* a job is compressed in chunks of 4 blocks, each its own frame-chunk call (:753) */
struct ZhipJob {
    uint32_t start;         // first byte of the section, from the frame start (the unit's srcLen is the section's length)
    uint32_t prefixLen;     // bytes in front of it the job's window covers (0 for the first job)
    uint32_t flags;
    uint32_t ownHeader;     // bytes of the frame header a later job writes and discards — they count in its `savings` (:737, zstd_compress.c:4538)
    uint64_t frameSize;     // content size of the whole frame (header)
    uint32_t frameIdx;      // which frame of the batch the job belongs to (its checksum, its size)
    uint32_t pad0;
};

// Where a frame's (or job's) hash table lives.  ZSTD_fast with hashLog <= 14: LDS — as 24-bit entries (Lds24Tab, 3 << hashLog bytes)
// when every position of the walk, counted from the start of its window, stays below 2^24, else as 32-bit words (4 << hashLog);
// larger tables and ZSTD_dfast's pair: HBM.  `span` = the bytes the workgroup walks over plus the prefix in front of them.
enum { ZHIP_FT_HBM = 0, ZHIP_FT_LDS24 = 1, ZHIP_FT_LDS32 = 2 };
__host__ __device__ inline uint32_t frame_table_mode(uint32_t strategy, uint32_t hashLog, uint64_t span)
{
    if (strategy != ZHIP_STRAT_FAST || hashLog > ZHIP_FRAME_LDS_HASHLOG) return ZHIP_FT_HBM;
    return span < ((uint64_t)1 << 24) ? ZHIP_FT_LDS24 : ZHIP_FT_LDS32;
}
__host__ __device__ inline uint32_t frame_table_lds_bytes(uint32_t mode, uint32_t hashLog)
{
    return mode == ZHIP_FT_LDS24 ? (3u << hashLog) : (mode == ZHIP_FT_LDS32 ? (4u << hashLog) : 0u);
}
__host__ __device__ inline uint32_t frame_lds_bytes(uint32_t tableBytes /* largest frame_table_lds_bytes of the launch */)
{
    uint32_t const base = (uint32_t)((sizeof(EntShared) + 15) & ~(size_t)15) + (uint32_t)sizeof(FrameShared);
    return base + ((tableBytes + 15u) & ~15u);
}
// words of table memory a frame needs: ZSTD_fast one table, ZSTD_dfast the long table followed by the short one
__host__ __device__ inline size_t frame_table_words(uint32_t strategy, uint32_t hashLog, uint32_t chainLog)
{
    return ((size_t)1 << hashLog) + (strategy == ZHIP_STRAT_DFAST ? (size_t)1 << chainLog : 0);
}

// ZSTD_writeFrameHeader (zstd_compress.c:4640-4690) with the content size known, no dictionary id
__host__ __device__ inline uint32_t frame_header_bytes_multi(uint32_t n, uint32_t windowLog)
{
    bool const single = ((uint64_t)1 << windowLog) >= n;
    uint32_t const fcs = (n >= 256) + (n >= 65536 + 256);
    return 4 + 1 + (single ? 0 : 1) + (fcs == 0 ? (single ? 1 : 0) : (fcs == 1 ? 2 : 4));
}
__device__ inline uint32_t write_frame_header_multi(uint8_t* op, uint32_t n, uint32_t windowLog, bool checksum)
{
    bool const single = ((uint64_t)1 << windowLog) >= n;
    uint32_t const fcs = (n >= 256) + (n >= 65536 + 256);
    op[0] = 0x28; op[1] = 0xB5; op[2] = 0x2F; op[3] = 0xFD;
    op[4] = (uint8_t)((checksum ? 4u : 0u) + ((uint32_t)single << 5) + (fcs << 6));
    uint32_t pos = 5;
    if (!single) op[pos++] = (uint8_t)((windowLog - 10) << 3);
    if (fcs == 0) { if (single) op[pos++] = (uint8_t)n; }
    else if (fcs == 1) { uint32_t const v = n - 256; op[pos++] = (uint8_t)v; op[pos++] = (uint8_t)(v >> 8); }
    else { op[pos++] = (uint8_t)n; op[pos++] = (uint8_t)(n >> 8); op[pos++] = (uint8_t)(n >> 16); op[pos++] = (uint8_t)(n >> 24); }
    return pos;
}

// the block parser as a called function: the frame kernel holds five instantiations of it next to the whole block encoder,
// and inlined they left the scalar register file with ~1000 spilled values
template <uint32_t MLS, int OCC>
__device__ __attribute__((noinline)) void parse_fast_block_far(const uint8_t* src, uint32_t b0, uint32_t n, uint32_t prefixLow, uint32_t maxRep,
                                                               uint32_t rep1, uint32_t rep2, uint32_t rep3, ZhipUnit u, WideTab T,
                                                               ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    parse_fast_block<MLS, WideTab>(src, b0, n, prefixLow, maxRep, rep1, rep2, rep3, u, T, seqs, lits, meta);
}

template <uint32_t MLS, int OCC>
__device__ __attribute__((noinline)) void parse_fast_block_far24(const uint8_t* src, uint32_t b0, uint32_t n, uint32_t prefixLow, uint32_t maxRep,
                                                                 uint32_t rep1, uint32_t rep2, uint32_t rep3, ZhipUnit u, Lds24Tab T,
                                                                 ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    parse_fast_block<MLS, Lds24Tab>(src, b0, n, prefixLow, maxRep, rep1, rep2, rep3, u, T, seqs, lits, meta);
}

template <uint32_t MLS, int OCC>
__device__ __attribute__((noinline)) void parse_dfast_block_far(const uint8_t* src, uint32_t b0, uint32_t n, uint32_t prefixLow, uint32_t maxRep,
                                                                uint32_t rep1, uint32_t rep2, uint32_t rep3, ZhipUnit u, unsigned char* scratch,
                                                                uint32_t* tabL, uint32_t* tabS, ZhipSeq* seqs, uint8_t* lits, ZhipParse* meta)
{
    parse_dfast_block<MLS, true>(src, b0, n, prefixLow, maxRep, rep1, rep2, rep3, u, scratch, tabL, tabS, seqs, lits, meta);
}

// ZSTD_fillHashTableForCCtx / ZSTD_fillDoubleHashTableForCCtx with ZSTD_dtlm_fast (zstd_fast.c:50-86, zstd_double_fast.c:56-90) over
// the prefix [p0, p1): every third position, later positions win — an atomic max, so the order of the lanes does not matter
__device__ inline void frame_fill_prefix(const uint8_t* __restrict__ src, const ZhipUnit& u, const WideTab& T, uint32_t p0, uint32_t p1)
{
    if (p1 - p0 <= 8) return;                                                // zstd_compress.c:4902
    uint32_t const big = u.hashLog > u.chainLog ? u.hashLog : u.chainLog;
    uint32_t const maxDict = 8u << (big < 28 ? big : 28);                    // :4889-4896
    if (p1 - p0 > maxDict) p0 = p1 - maxDict;
    uint32_t const stop = p1 - 8 + 2;                                        // ip + 3 < iend + 2
    for (uint32_t ip = p0 + 3u * threadIdx.x; ip + 3 < stop; ip += 3u * ZHIP_ENT_THREADS) {
        uint64_t const b = ld64(src + ip);
        if (u.strategy == ZHIP_STRAT_DFAST) {
            uint32_t const hl = mulhi64_top32(b, 0xCF1BBCDCB7A56463ULL) >> (32 - u.hashLog);
            uint32_t hs;
            switch (u.minMatch) { case 5: hs = hash_pos<5>(b, 32 - u.chainLog); break; case 6: hs = hash_pos<6>(b, 32 - u.chainLog); break;
                                  case 7: hs = hash_pos<7>(b, 32 - u.chainLog); break; case 8: hs = hash_pos<8>(b, 32 - u.chainLog); break;
                                  default: hs = hash_pos<4>(b, 32 - u.chainLog); break; }
            atomicMax(&T.w[hl], ip);
            atomicMax(&T.w[((size_t)1 << u.hashLog) + hs], ip);
        } else {
            uint32_t h;
            switch (u.minMatch) { case 5: h = hash_pos<5>(b, 32 - u.hashLog); break; case 6: h = hash_pos<6>(b, 32 - u.hashLog); break;
                                  case 7: h = hash_pos<7>(b, 32 - u.hashLog); break; case 8: h = hash_pos<8>(b, 32 - u.hashLog); break;
                                  default: h = hash_pos<4>(b, 32 - u.hashLog); break; }
            atomicMax(&T.w[h], ip);
        }
    }
}

// the same fill for a 24-bit LDS table (ZSTD_fast only).  No 24-bit atomic exists, so the prefix is walked in pieces that do not cross
// a multiple of 64 KB: inside a piece every writer has the same hi byte.  Phase one: every writer stores its entry — whatever earlier
// pieces left in the slot is gone, as it would be in the serial loop.  Phase two: a 16-bit atomic max on lo[] (compare-and-swap on the
// 32-bit word that holds the half) leaves the piece's largest position.  Two barriers per piece, a trip count that does not depend
// on the data.  (A "store again until nobody lost" loop is NOT safe here: its exit test reads LDS, the compiler cannot see that it is
// uniform, restructures the loop per lane — and the lane that resets the flag sits masked while its wave runs the next barrier.)
__device__ inline void frame_fill_prefix24(const uint8_t* __restrict__ src, const ZhipUnit& u, const Lds24Tab& T, uint32_t* words /* lo[] as 32-bit words */,
                                           uint32_t p0, uint32_t p1)
{
    if (p1 - p0 <= 8) return;
    uint32_t const big = u.hashLog > u.chainLog ? u.hashLog : u.chainLog;        // the chain log counts even though ZSTD_fast has no chain (zstd_compress.c:4889-4896)
    uint32_t const maxDict = 8u << (big < 28 ? big : 28);
    if (p1 - p0 > maxDict) p0 = p1 - maxDict;
    uint32_t const stop = p1 - 8 + 2;
    for (uint32_t base = p0; base + 3 < stop; ) {
        uint32_t const segEnd = (base | 0xFFFFu) + 1u;                                    // first position of the next 64 KB segment
        uint32_t const ip = base + 3u * threadIdx.x;
        bool const mine = ip + 3 < stop && ip < segEnd;
        uint32_t h = 0;
        if (mine) {
            uint64_t const b = ld64(src + ip);
            switch (u.minMatch) { case 5: h = hash_pos<5>(b, 32 - u.hashLog); break; case 6: h = hash_pos<6>(b, 32 - u.hashLog); break;
                                  case 7: h = hash_pos<7>(b, 32 - u.hashLog); break; case 8: h = hash_pos<8>(b, 32 - u.hashLog); break;
                                  default: h = hash_pos<4>(b, 32 - u.hashLog); break; }
            tab_put(T, h, ip);
        }
        __syncthreads();
        if (mine) {
            uint32_t* const w = words + (h >> 1);
            uint32_t const sh = (h & 1u) * 16u, lo16 = ip & 0xFFFFu;
            uint32_t old = atomicOr(w, 0u);
            while (((old >> sh) & 0xFFFFu) < lo16) {
                uint32_t const got = atomicCAS(w, old, (old & ~(0xFFFFu << sh)) | (lo16 << sh));
                if (got == old) break;
                old = got;
            }
        }
        __syncthreads();
        uint32_t const next = base + 3u * ZHIP_ENT_THREADS;
        base = next < segEnd ? next : base + 3u * ((segEnd - base + 2u) / 3u);            // stay on the every-third grid across the border
    }
}

// Does a job count its positions from 1 (src one byte before its window)?  Not the frame's first job (a frame never inserts its
// position 0, like the reference); every other job does.  When its window starts at the frame's byte 0 (jobSize <= overlap, second
// job) the byte in front of the window does not exist: the frame parsers never touch position 0 (tab_guard; the ZSTD_dfast parser
// only reads the candidates of non-empty entries).
__host__ __device__ inline uint32_t frame_job_shift(const ZhipJob* job, uint32_t /*tableMode*/)
{
    return (!job || (job->flags & ZHIP_JOB_FIRST)) ? 0u : 1u;
}

// job == nullptr: the whole input src[0, u.srcLen) as ONE frame (ZSTD_compress2 without workers: one context, one frame chunk).
// job != nullptr: one job of a frame (see ZhipJob); src is the start of the job's WINDOW (its prefix) — one byte before it for a job
// that is not the frame's first — and positions count from there.
// use24: the table is T24 (LDS, 24-bit entries), else T.
template <int OCC /* the kernel's waves per SIMD: its callees are instantiated per kernel so that each gets that kernel's register budget */>
__device__ inline void frame_fast(const uint8_t* __restrict__ src, const ZhipUnit& u, const WideTab& T, const Lds24Tab& T24, bool use24, ZhipSeq* seqs, uint8_t* lits,
                                  uint16_t* stBits, uint32_t seqCap, uint8_t* __restrict__ out, uint32_t* outSize,
                                  EntShared* sh, FrameShared* fs, ZhipFrameState* st, bool withChecksum, uint32_t checksum,
                                  const ZhipJob* __restrict__ job, uint32_t winStart /* 0, or 1 for a job whose src is one byte before its window */)
{
    int const t = (int)threadIdx.x, wv = t >> 6;
    bool const first = !job || (job->flags & ZHIP_JOB_FIRST), lastJob = !job || (job->flags & ZHIP_JOB_LAST);
    // A table entry of 0 means "empty", and a frame never inserts its position 0 (zstd_fast.c:238).  A later job's prefix starts with
    // a position the reference CAN match (its indices start at 2), so such a job counts from winStart = 1: src points one byte before
    // its window (the caller decides, see frame_job_shift).
    uint32_t const j0 = winStart + (job ? job->prefixLen : 0u);              // the section [j0, jEnd) behind the prefix [winStart, j0)
    uint32_t const n = u.srcLen, jEnd = j0 + n;
    uint32_t const frameSize = job ? (uint32_t)job->frameSize : n;
    uint32_t op = first ? frame_header_bytes_multi(frameSize, u.windowLog) : 0u;
    if (first && t == 0) write_frame_header_multi(out, frameSize, u.windowLog, withChecksum);
    if (n == 0) {                                                            // :5270 an empty frame is one empty raw block
        if (t == 0) {
            out[op] = 1; out[op + 1] = 0; out[op + 2] = 0; op += 3;
            if (withChecksum) { for (int b = 0; b < 4; b++) out[op + b] = (uint8_t)(checksum >> (8 * b)); op += 4; }
            *outSize = op;
        }
        return;
    }
    if (use24) { lds_u32* const z = (lds_u32*)T24.lo; for (uint32_t i = (uint32_t)t; i < (3u << u.hashLog) >> 2; i += ZHIP_ENT_THREADS) z[i] = 0; }   // lo[] and hi[] are contiguous
    else for (uint32_t i = (uint32_t)t; i < (uint32_t)frame_table_words(u.strategy, u.hashLog, u.chainLog); i += ZHIP_ENT_THREADS) T.w[i] = 0;     // fresh table(s) (:2020)
    if (t == 0) { st->ent.hufRepeat = 0; st->ent.hufMaxSym = 0; st->ent.fseRepeat[0] = 0; st->ent.fseRepeat[1] = 0; st->ent.fseRepeat[2] = 0; }
    __syncthreads();
    if (job && job->prefixLen) { if (use24) frame_fill_prefix24(src, u, T24, T.w, winStart, j0); else frame_fill_prefix(src, u, T, winStart, j0); __syncthreads(); }
    uint32_t rep1 = first ? 1u : 0u, rep2 = first ? 4u : 0u, rep3 = first ? 8u : 0u;     // later jobs: ZSTD_invalidateRepCodes (:741)
    long long savings = (job && !first) ? -(long long)job->ownHeader : 0;
    uint32_t pos = j0;
    uint32_t const maxDist = 1u << u.windowLog;
    __syncthreads();
    while (pos < jEnd) {
        // a job hands its section to the compressor in chunks of 512 KB (one ZSTD_compressContinue each): the block rule sees what
        // is left of the CHUNK; without jobs the whole input is one chunk
        uint32_t const chunkEnd = job ? (jEnd - pos > ZHIP_JOB_CHUNK - ((pos - j0) & (ZHIP_JOB_CHUNK - 1)) ? pos + ZHIP_JOB_CHUNK - ((pos - j0) & (ZHIP_JOB_CHUNK - 1)) : jEnd) : jEnd;
        uint32_t const remaining = chunkEnd - pos;
        uint32_t bLen = remaining < ZHIP_UNIT_MAX ? remaining : ZHIP_UNIT_MAX;                      // :4494-4518
        if (remaining >= ZHIP_UNIT_MAX && savings >= 3) bLen = ZHIP_FRAME_BLOCK_SPLIT;
        uint32_t const last = (lastJob && pos + bLen == jEnd) ? 1u : 0u;
        uint32_t const end = pos + bLen;
        uint8_t* const body = out + op + 3;
        uint32_t cSize = 0;
        if (bLen >= 7) {                                                                         // :3216
            if (wv == 0) {
                // ZSTD_window_enforceMaxDist from the block start (:4555), then ZSTD_getLowestPrefixIndex(block end) (zstd_fast.c:205)
                uint32_t const dl0 = pos > maxDist ? pos - maxDist : 0, dictLimit = dl0 > winStart ? dl0 : winStart;
                uint32_t const prefixLow = (end - dictLimit > maxDist) ? end - maxDist : dictLimit;
                uint32_t const ip0 = pos + (pos == prefixLow);
                uint32_t const windowLow = (ip0 - dictLimit > maxDist) ? ip0 - maxDist : dictLimit;
                uint32_t const maxRep = ip0 - windowLow;
                if (u.strategy == ZHIP_STRAT_DFAST) {
                    uint32_t* const tL = T.w; uint32_t* const tS = T.w + ((size_t)1 << u.hashLog);
                    switch (u.minMatch) {
                    case 5:  parse_dfast_block_far<5, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, fs->dfScratch, tL, tS, seqs, lits, &fs->meta); break;
                    case 6:  parse_dfast_block_far<6, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, fs->dfScratch, tL, tS, seqs, lits, &fs->meta); break;
                    case 7:  parse_dfast_block_far<7, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, fs->dfScratch, tL, tS, seqs, lits, &fs->meta); break;
                    case 8:  parse_dfast_block_far<8, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, fs->dfScratch, tL, tS, seqs, lits, &fs->meta); break;
                    default: parse_dfast_block_far<4, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, fs->dfScratch, tL, tS, seqs, lits, &fs->meta); break;
                    }
                } else if (use24)
                switch (u.minMatch) {
                case 5:  parse_fast_block_far24<5, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T24, seqs, lits, &fs->meta); break;
                case 6:  parse_fast_block_far24<6, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T24, seqs, lits, &fs->meta); break;
                case 7:  parse_fast_block_far24<7, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T24, seqs, lits, &fs->meta); break;
                case 8:  parse_fast_block_far24<8, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T24, seqs, lits, &fs->meta); break;
                default: parse_fast_block_far24<4, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T24, seqs, lits, &fs->meta); break;
                }
                else
                switch (u.minMatch) {                                        // the hash width is a compile-time constant inside the parser
                case 5:  parse_fast_block_far<5, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T, seqs, lits, &fs->meta); break;
                case 6:  parse_fast_block_far<6, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T, seqs, lits, &fs->meta); break;
                case 7:  parse_fast_block_far<7, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T, seqs, lits, &fs->meta); break;
                case 8:  parse_fast_block_far<8, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T, seqs, lits, &fs->meta); break;
                default: parse_fast_block_far<4, OCC>(src, pos, end, prefixLow, maxRep, rep1, rep2, rep3, u, T, seqs, lits, &fs->meta); break;
                }
            }
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
                if (sh->litMode == 2 && sh->litType == 2) {                  // a new Huffman table (zstd_compress_literals.c:223-226)
                    st->ent.hufCode[t] = sh->code[t];
                    if (t == 0) { st->ent.hufRepeat = 1; st->ent.hufMaxSym = sh->hufMaxSym; }
                }
            }
        }
        uint32_t total;
        if (cSize == 0) {                                                    // :4592 ZSTD_noCompressBlock
            __syncthreads();                                                 // the encoder's stray header bytes land first
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
        // the first job wrote the real frame header in its first call: from the second chunk on it counts as produced (zstd_compress.c:4767)
        if (job && first && end == chunkEnd && end - j0 <= ZHIP_JOB_CHUNK) savings -= (long long)frame_header_bytes_multi(frameSize, u.windowLog);
        pos = end;
        __syncthreads();                                                     // the block's bytes and the new state are in place
    }
    if (t == 0) {
        if (withChecksum && lastJob) { for (int b = 0; b < 4; b++) out[op + b] = (uint8_t)(checksum >> (8 * b)); op += 4; }   // :5297-5303, zstdmt_compress.c:1516
        *outSize = op;
    }
}

}  // namespace zhip
