/* oracle/ref_shim.c — TEST INFRASTRUCTURE ONLY.
 *
 * Thin driver around the REAL reference (facebook/zstd, compiled from /root/reference into
 * oracle/_ref/libzstd_ref.so by oracle/Makefile).  It exposes, with a flat C ABI that python/ctypes
 * can call, exactly the reference entry points SURVEY.md §8(c) names as oracles:
 *
 *   O1  zref_compress_chunks    = ZSTD_compress2 per chunk on one reused CCtx   (what `zstd -b# -B<chunk>` times,
 *                                 programs/benchzstd.c:336-345)
 *   O2  zref_sequences          = ZSTD_generateSequences on a DEDICATED CCtx    (lib/compress/zstd_compress.c:3462;
 *                                 the call poisons its CCtx)
 *   O3  zref_huf_*, zref_fse_*  = stage functions exported by the static lib    (huf_compress.c, fse_compress.c)
 *   O4  zref_decompress         = ZSTD_decompress                               (lib/decompress/zstd_decompress.c:1201)
 *       zref_datagen / zref_datagen_stream / zref_lorem = programs/datagen.c, programs/lorem.c input generators
 *       zref_get_cparams        = ZSTD_getCParams                               (lib/compress/zstd_compress.c:7150)
 *
 * Nothing here is linked by, imported by, or shipped with the product library.
 */
#define ZSTD_STATIC_LINKING_ONLY
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "zstd.h"
#include "zstd_errors.h"
#include "datagen.h"
#include "lorem.h"
#include "hist.h"
#include "huf.h"
#include "fse.h"
#include "zdict.h"
#include "zstd_seekable.h"

static void set_level(ZSTD_CCtx* c, int level)
{
    ZSTD_CCtx_reset(c, ZSTD_reset_session_and_parameters);
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
}

/* O1: one frame per chunk; returns total bytes or (size_t)-1. sizes[i] = compressed size of chunk i. */
size_t zref_compress_chunks(int level, size_t chunkSize, const void* src, size_t n,
                            void* dst, size_t dstCap, size_t* sizes, size_t maxChunks)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t pos = 0, off = 0, k = 0;
    if (!c) return (size_t)-1;
    set_level(c, level);
    if (n == 0) {
        size_t r = ZSTD_compress2(c, dst, dstCap, src, 0);
        ZSTD_freeCCtx(c);
        if (ZSTD_isError(r)) return (size_t)-1;
        if (sizes && maxChunks) sizes[0] = r;
        return r;
    }
    while (off < n) {
        size_t const len = (n - off < chunkSize) ? n - off : chunkSize;
        size_t const r = ZSTD_compress2(c, (char*)dst + pos, dstCap - pos, (const char*)src + off, len);
        if (ZSTD_isError(r)) { ZSTD_freeCCtx(c); return (size_t)-1; }
        if (sizes && k < maxChunks) sizes[k] = r;
        k++; pos += r; off += len;
    }
    ZSTD_freeCCtx(c);
    return pos;
}

/* O1 for the hash-chain strategies: as above with ZSTD_c_useRowMatchFinder = ZSTD_ps_disable, so that greedy / lazy /
 * lazy2 use ZSTD_HcFindBestMatch (zstd_lazy.c:667) instead of the row-hash matcher. */
size_t zref_compress_chunks_norow(int level, size_t chunkSize, const void* src, size_t n,
                                  void* dst, size_t dstCap, size_t* sizes, size_t maxChunks)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t pos = 0, off = 0, k = 0;
    if (!c) return (size_t)-1;
    set_level(c, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_useRowMatchFinder, ZSTD_ps_disable);
    if (n == 0) {
        size_t r = ZSTD_compress2(c, dst, dstCap, src, 0);
        ZSTD_freeCCtx(c);
        if (ZSTD_isError(r)) return (size_t)-1;
        if (sizes && maxChunks) sizes[0] = r;
        return r;
    }
    while (off < n) {
        size_t const len = (n - off < chunkSize) ? n - off : chunkSize;
        size_t const r = ZSTD_compress2(c, (char*)dst + pos, dstCap - pos, (const char*)src + off, len);
        if (ZSTD_isError(r)) { ZSTD_freeCCtx(c); return (size_t)-1; }
        if (sizes && k < maxChunks) sizes[k] = r;
        k++; pos += r; off += len;
    }
    ZSTD_freeCCtx(c);
    return pos;
}

/* Same but with every cParam pinned explicitly (windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy). */
size_t zref_compress_chunks_params(const int cp[7], size_t chunkSize, const void* src, size_t n,
                                   void* dst, size_t dstCap, size_t* sizes, size_t maxChunks)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t pos = 0, off = 0, k = 0;
    if (!c) return (size_t)-1;
    ZSTD_CCtx_setParameter(c, ZSTD_c_windowLog, cp[0]);
    ZSTD_CCtx_setParameter(c, ZSTD_c_chainLog, cp[1]);
    ZSTD_CCtx_setParameter(c, ZSTD_c_hashLog, cp[2]);
    ZSTD_CCtx_setParameter(c, ZSTD_c_searchLog, cp[3]);
    ZSTD_CCtx_setParameter(c, ZSTD_c_minMatch, cp[4]);
    ZSTD_CCtx_setParameter(c, ZSTD_c_targetLength, cp[5]);
    ZSTD_CCtx_setParameter(c, ZSTD_c_strategy, cp[6]);
    while (off < n) {
        size_t const len = (n - off < chunkSize) ? n - off : chunkSize;
        size_t const r = ZSTD_compress2(c, (char*)dst + pos, dstCap - pos, (const char*)src + off, len);
        if (ZSTD_isError(r)) { ZSTD_freeCCtx(c); return (size_t)-1; }
        if (sizes && k < maxChunks) sizes[k] = r;
        k++; pos += r; off += len;
    }
    ZSTD_freeCCtx(c);
    return pos;
}

/* a level plus explicitly set parameters (0 = leave the level's), optionally with the row matcher disabled */
size_t zref_compress_chunks_level_params(int level, const int cp[7], int noRow, size_t chunkSize, const void* src, size_t n, void* dst, size_t dstCap)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t pos = 0, off = 0;
    static const ZSTD_cParameter ids[7] = { ZSTD_c_windowLog, ZSTD_c_chainLog, ZSTD_c_hashLog, ZSTD_c_searchLog, ZSTD_c_minMatch, ZSTD_c_targetLength, ZSTD_c_strategy };
    int i;
    if (!c) return (size_t)-1;
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
    for (i = 0; i < 7; i++) if (cp[i]) { if (ZSTD_isError(ZSTD_CCtx_setParameter(c, ids[i], cp[i]))) { ZSTD_freeCCtx(c); return (size_t)-1; } }
    if (noRow) ZSTD_CCtx_setParameter(c, ZSTD_c_useRowMatchFinder, ZSTD_ps_disable);
    do {
        size_t const len = (n - off < chunkSize) ? n - off : chunkSize;
        size_t const r = ZSTD_compress2(c, (char*)dst + pos, dstCap - pos, (const char*)src + off, len);
        if (ZSTD_isError(r)) { ZSTD_freeCCtx(c); return (size_t)-1; }
        pos += r; off += len;
    } while (off < n);
    ZSTD_freeCCtx(c);
    return pos;
}

/* records[] (sizes in recSizes[nRec], laid out back to back in src) each compressed as its own frame with a CDict made from
 * `dict` at `level` — ZSTD_createCDict + ZSTD_CCtx_refCDict + ZSTD_compress2 (the contrib/largeNbDicts / `zstd -D` shape).
 * outSizes[nRec] receives the frame sizes; returns the total, (size_t)-1 on error. */
size_t zref_compress_records_cdict(int level, const void* dict, size_t dictSize, const void* src, const size_t* recSizes, size_t nRec,
                                   void* dst, size_t dstCap, size_t* outSizes)
{
    ZSTD_CDict* cd = ZSTD_createCDict(dict, dictSize, level);
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t pos = 0, off = 0, k;
    if (!cd || !c) { ZSTD_freeCDict(cd); ZSTD_freeCCtx(c); return (size_t)-1; }
    if (ZSTD_isError(ZSTD_CCtx_refCDict(c, cd))) { ZSTD_freeCDict(cd); ZSTD_freeCCtx(c); return (size_t)-1; }
    for (k = 0; k < nRec; k++) {
        size_t const r = ZSTD_compress2(c, (char*)dst + pos, dstCap - pos, (const char*)src + off, recSizes[k]);
        if (ZSTD_isError(r)) { ZSTD_freeCDict(cd); ZSTD_freeCCtx(c); return (size_t)-1; }
        if (outSizes) outSizes[k] = r;
        pos += r; off += recSizes[k];
    }
    ZSTD_freeCDict(cd); ZSTD_freeCCtx(c);
    return pos;
}

/* the same with a FRESH CCtx per record (the row matcher's hash salt depends on what a context compressed before) and, optionally, the
 * row matcher disabled for CDict and CCtx alike (ZSTD_createCDict_advanced2 with ZSTD_c_useRowMatchFinder = ZSTD_ps_disable) */
size_t zref_compress_records_cdict_fresh(int level, int noRow, const void* dict, size_t dictSize, const void* src, const size_t* recSizes, size_t nRec,
                                         void* dst, size_t dstCap, size_t* outSizes)
{
    ZSTD_CCtx_params* p = ZSTD_createCCtxParams();
    ZSTD_CDict* cd;
    size_t pos = 0, off = 0, k;
    ZSTD_customMem const mem = { NULL, NULL, NULL };
    if (!p) return (size_t)-1;
    ZSTD_CCtxParams_init(p, level);
    if (noRow) ZSTD_CCtxParams_setParameter(p, ZSTD_c_useRowMatchFinder, ZSTD_ps_disable);
    cd = ZSTD_createCDict_advanced2(dict, dictSize, ZSTD_dlm_byCopy, ZSTD_dct_auto, p, mem);
    ZSTD_freeCCtxParams(p);
    if (!cd) return (size_t)-1;
    for (k = 0; k < nRec; k++) {
        ZSTD_CCtx* c = ZSTD_createCCtx();
        size_t r;
        if (!c) { ZSTD_freeCDict(cd); return (size_t)-1; }
        if (noRow) ZSTD_CCtx_setParameter(c, ZSTD_c_useRowMatchFinder, ZSTD_ps_disable);
        ZSTD_CCtx_refCDict(c, cd);
        r = ZSTD_compress2(c, (char*)dst + pos, dstCap - pos, (const char*)src + off, recSizes[k]);
        ZSTD_freeCCtx(c);
        if (ZSTD_isError(r)) { ZSTD_freeCDict(cd); return (size_t)-1; }
        if (outSizes) outSizes[k] = r;
        pos += r; off += recSizes[k];
    }
    ZSTD_freeCDict(cd);
    return pos;
}

/* decode one frame with a dictionary (validator) */
size_t zref_decompress_dict(void* dst, size_t cap, const void* src, size_t n, const void* dict, size_t dictSize)
{
    ZSTD_DCtx* d = ZSTD_createDCtx();
    size_t const r = ZSTD_decompress_usingDict(d, dst, cap, src, n, dict, dictSize);
    ZSTD_freeDCtx(d);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

/* ZDICT_trainFromBuffer (lib/zdict.h:210): the dictionary `zstd --train` builds; returns its size or (size_t)-1 */
size_t zref_train_dict(void* dictBuf, size_t dictCap, const void* samples, const size_t* sampleSizes, unsigned nbSamples)
{
    size_t const r = ZDICT_trainFromBuffer(dictBuf, dictCap, samples, sampleSizes, nbSamples);
    return ZDICT_isError(r) ? (size_t)-1 : r;
}

/* O1 with ZSTD_c_checksumFlag = 1 (what the zstd CLI does by default, programs/fileio.c:287) */
size_t zref_compress_chunks_checksum(int level, size_t chunkSize, const void* src, size_t n, void* dst, size_t dstCap, size_t* sizes, size_t maxChunks)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t pos = 0, off = 0, k = 0;
    if (!c) return (size_t)-1;
    set_level(c, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, 1);
    do {
        size_t const len = (n - off < chunkSize) ? n - off : chunkSize;
        size_t const r = ZSTD_compress2(c, (char*)dst + pos, dstCap - pos, (const char*)src + off, len);
        if (ZSTD_isError(r)) { ZSTD_freeCCtx(c); return (size_t)-1; }
        if (sizes && k < maxChunks) sizes[k] = r;
        k++; pos += r; off += len;
    } while (off < n);
    ZSTD_freeCCtx(c);
    return pos;
}

/* contrib/seekable_format: the reference's own seek-table writer (frame log API) and seekable decoder */
size_t zref_seek_table(void* dst, size_t cap, const unsigned* cs, const unsigned* ds, const unsigned* ck, unsigned n)
{
    ZSTD_frameLog* fl = ZSTD_seekable_createFrameLog(ck != NULL);
    ZSTD_outBuffer out = { dst, cap, 0 };
    unsigned i; size_t r;
    for (i = 0; i < n; i++) ZSTD_seekable_logFrame(fl, cs[i], ds[i], ck ? ck[i] : 0);
    r = ZSTD_seekable_writeSeekTable(fl, &out);
    ZSTD_seekable_freeFrameLog(fl);
    return (ZSTD_isError(r) || r != 0) ? (size_t)-1 : out.pos;
}
size_t zref_seekable_read(void* dst, size_t len, const void* src, size_t n, unsigned long long offset, unsigned* nFrames)
{
    ZSTD_seekable* zs = ZSTD_seekable_create();
    size_t r = ZSTD_seekable_initBuff(zs, src, n);
    if (ZSTD_isError(r)) { ZSTD_seekable_free(zs); return (size_t)-1; }
    if (nFrames) *nFrames = ZSTD_seekable_getNumFrames(zs);
    r = ZSTD_seekable_decompress(zs, dst, len, offset);
    ZSTD_seekable_free(zs);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

/* whole buffer as ONE frame (the conventional `zstd -b#` figure; NOT the parity target) */
size_t zref_compress_frame(int level, const void* src, size_t n, void* dst, size_t dstCap)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r;
    if (!c) return (size_t)-1;
    set_level(c, level);
    r = ZSTD_compress2(c, dst, dstCap, src, n);
    ZSTD_freeCCtx(c);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

/* ZSTD_compress2 with ZSTD_c_nbWorkers = 1 (zstdmt_compress.c): the input is cut into jobs of jobSize bytes (0 = the default,
 * 1 << max(20, windowLog + 2)), every job after the first starts from a fresh context that has only loaded the last `overlap`
 * bytes of the previous job as a prefix; the jobs' blocks are emitted back to back inside ONE frame.  overlapLog 0 = default. */
size_t zref_compress_frame_mt(int level, const int cp[7], unsigned long long jobSize, int overlapLog, int checksumFlag,
                              const void* src, size_t n, void* dst, size_t dstCap)
{
    static const ZSTD_cParameter ids[7] = { ZSTD_c_windowLog, ZSTD_c_chainLog, ZSTD_c_hashLog, ZSTD_c_searchLog, ZSTD_c_minMatch, ZSTD_c_targetLength, ZSTD_c_strategy };
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r; int i;
    if (!c) return (size_t)-1;
    set_level(c, level);
    if (cp) for (i = 0; i < 7; i++) if (cp[i]) ZSTD_CCtx_setParameter(c, ids[i], cp[i]);
    if (ZSTD_isError(ZSTD_CCtx_setParameter(c, ZSTD_c_nbWorkers, 1))) { ZSTD_freeCCtx(c); return (size_t)-1; }
    if (jobSize) ZSTD_CCtx_setParameter(c, ZSTD_c_jobSize, (int)jobSize);
    if (overlapLog) ZSTD_CCtx_setParameter(c, ZSTD_c_overlapLog, overlapLog);
    ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, checksumFlag);
    r = ZSTD_compress2(c, dst, dstCap, src, n);
    ZSTD_freeCCtx(c);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

/* the same with the row matcher switched off (ZSTD_c_useRowMatchFinder = ZSTD_ps_disable): the lazy strategies' hash chain */
size_t zref_compress_frame_mt_norow(int level, const int cp[7], unsigned long long jobSize, int overlapLog, int checksumFlag, int noRow,
                                    const void* src, size_t n, void* dst, size_t dstCap)
{
    static const ZSTD_cParameter ids[7] = { ZSTD_c_windowLog, ZSTD_c_chainLog, ZSTD_c_hashLog, ZSTD_c_searchLog, ZSTD_c_minMatch, ZSTD_c_targetLength, ZSTD_c_strategy };
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r; int i;
    if (!c) return (size_t)-1;
    set_level(c, level);
    if (cp) for (i = 0; i < 7; i++) if (cp[i]) ZSTD_CCtx_setParameter(c, ids[i], cp[i]);
    if (noRow) ZSTD_CCtx_setParameter(c, ZSTD_c_useRowMatchFinder, ZSTD_ps_disable);
    if (ZSTD_isError(ZSTD_CCtx_setParameter(c, ZSTD_c_nbWorkers, 1))) { ZSTD_freeCCtx(c); return (size_t)-1; }
    if (jobSize) ZSTD_CCtx_setParameter(c, ZSTD_c_jobSize, (int)jobSize);
    if (overlapLog) ZSTD_CCtx_setParameter(c, ZSTD_c_overlapLog, overlapLog);
    ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, checksumFlag);
    r = ZSTD_compress2(c, dst, dstCap, src, n);
    ZSTD_freeCCtx(c);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

/* one frame with chosen frame parameters: contentSizeFlag (0 = the header does not state the size, what streaming without a
 * pledged size emits), checksumFlag, windowLog (0 = level default) — for the decoder tests */
size_t zref_compress_frame_params(int level, int contentSizeFlag, int checksumFlag, int windowLog, const void* src, size_t n, void* dst, size_t dstCap)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r;
    if (!c) return (size_t)-1;
    set_level(c, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_contentSizeFlag, contentSizeFlag);
    ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, checksumFlag);
    if (windowLog) ZSTD_CCtx_setParameter(c, ZSTD_c_windowLog, windowLog);
    r = ZSTD_compress2(c, dst, dstCap, src, n);
    ZSTD_freeCCtx(c);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

/* the same with the reference's DEFAULT matcher selection (row hash for greedy / lazy / lazy2 when windowLog > 14) on a fresh CCtx */
size_t zref_sequences_default(int level, const void* src, size_t n, unsigned* out, size_t capSeqs)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    ZSTD_Sequence* s = (ZSTD_Sequence*)malloc(sizeof(ZSTD_Sequence) * capSeqs);
    size_t r, i;
    if (!c || !s) return (size_t)-1;
    set_level(c, level);
    r = ZSTD_generateSequences(c, s, capSeqs, src, n);
    if (ZSTD_isError(r)) { free(s); ZSTD_freeCCtx(c); return (size_t)-1; }
    for (i = 0; i < r; i++) { out[4*i+0] = s[i].offset; out[4*i+1] = s[i].litLength; out[4*i+2] = s[i].matchLength; out[4*i+3] = s[i].rep; }
    free(s); ZSTD_freeCCtx(c);
    return r;
}

/* O2: sequences of ONE unit (n <= 128 KB) from the internal block compressor. out = 4 u32 per sequence
 * {offset, litLength, matchLength, rep}; block delimiter {0,lastLits,0,0} included. */
size_t zref_sequences(int level, const void* src, size_t n, unsigned* out, size_t capSeqs)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();            /* dedicated: generateSequences poisons it */
    ZSTD_Sequence* s = (ZSTD_Sequence*)malloc(sizeof(ZSTD_Sequence) * capSeqs);
    size_t r, i;
    if (!c || !s) return (size_t)-1;
    set_level(c, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_useRowMatchFinder, ZSTD_ps_disable);   /* no effect below greedy */
    r = ZSTD_generateSequences(c, s, capSeqs, src, n);
    if (ZSTD_isError(r)) { free(s); ZSTD_freeCCtx(c); return (size_t)-1; }
    for (i = 0; i < r; i++) {
        out[4*i+0] = s[i].offset; out[4*i+1] = s[i].litLength;
        out[4*i+2] = s[i].matchLength; out[4*i+3] = s[i].rep;
    }
    free(s); ZSTD_freeCCtx(c);
    return r;
}

/* sequences of ONE source compressed with a CDict attached (dedicated CCtx, see N5) */
size_t zref_sequences_cdict(int level, const void* dict, size_t dictSize, const void* src, size_t n, unsigned* out, size_t capSeqs)
{
    ZSTD_CDict* cd = ZSTD_createCDict(dict, dictSize, level);
    ZSTD_CCtx* c = ZSTD_createCCtx();
    ZSTD_Sequence* s = (ZSTD_Sequence*)malloc(sizeof(ZSTD_Sequence) * capSeqs);
    size_t r, i;
    if (!c || !s || !cd) return (size_t)-1;
    ZSTD_CCtx_refCDict(c, cd);
    r = ZSTD_generateSequences(c, s, capSeqs, src, n);
    if (ZSTD_isError(r)) { free(s); ZSTD_freeCCtx(c); ZSTD_freeCDict(cd); return (size_t)-1; }
    for (i = 0; i < r; i++) {
        out[4*i+0] = s[i].offset; out[4*i+1] = s[i].litLength;
        out[4*i+2] = s[i].matchLength; out[4*i+3] = s[i].rep;
    }
    free(s); ZSTD_freeCCtx(c); ZSTD_freeCDict(cd);
    return r;
}

size_t zref_decompress(void* dst, size_t dstCap, const void* src, size_t n)
{
    size_t const r = ZSTD_decompress(dst, dstCap, src, n);
    return ZSTD_isError(r) ? (size_t)-1 : r;
}

unsigned long long zref_decompressed_size(const void* src, size_t n)
{
    return ZSTD_findDecompressedSize(src, n);
}

size_t zref_compress_bound(size_t n) { return ZSTD_compressBound(n); }

void zref_get_cparams(int level, unsigned long long srcSize, size_t dictSize, int out[7])
{
    ZSTD_compressionParameters const p = ZSTD_getCParams(level, srcSize, dictSize);
    out[0] = (int)p.windowLog; out[1] = (int)p.chainLog; out[2] = (int)p.hashLog; out[3] = (int)p.searchLog;
    out[4] = (int)p.minMatch;  out[5] = (int)p.targetLength; out[6] = (int)p.strategy;
}

/* ---- input generators ---- */
void zref_datagen(void* buf, size_t size, double matchProba, double litProba, unsigned seed)
{
    RDG_genBuffer(buf, size, matchProba, litProba, seed);
}
void zref_lorem(void* buf, size_t size, unsigned seed) { LOREM_genBuffer(buf, size, seed); }

/* ---- O3 stage oracles ---- */
/* histogram: returns largest count, writes count[256] and *maxSym */
size_t zref_hist(unsigned* count, unsigned* maxSym, const void* src, size_t n)
{
    *maxSym = 255;
    return HIST_count(count, maxSym, src, n);
}

/* Huffman code lengths the reference assigns for a histogram (HUF_buildCTable_wksp, huf_compress.c:756).
 * nbBits[s] for s<=maxSym; returns tableLog (maxNbBits) or (size_t)-1 */
size_t zref_huf_build(const unsigned* count, unsigned maxSym, unsigned maxNbBits, unsigned char* nbBits)
{
    HUF_CREATE_STATIC_CTABLE(ct, 255);
    static unsigned wksp[HUF_WORKSPACE_SIZE_U64 * 2];
    unsigned s;
    size_t const r = HUF_buildCTable_wksp(ct, count, maxSym, maxNbBits, wksp, sizeof(wksp));
    if (HUF_isError(r)) return (size_t)-1;
    for (s = 0; s <= maxSym; s++) nbBits[s] = (unsigned char)HUF_getNbBitsFromCTable(ct, s);
    return r;
}

/* Full literals-section body through HUF_compress4X_repeat / 1X with no previous table
 * (huf_compress.c:1453): returns size, 0 (not compressible), 1 (rle) */
size_t zref_huf_compress(void* dst, size_t dstCap, const void* src, size_t n, int fourStreams, int flags)
{
    static unsigned long long wksp[HUF_WORKSPACE_SIZE_U64];
    HUF_CREATE_STATIC_CTABLE(ct, 255);
    HUF_repeat rep = HUF_repeat_none;
    size_t r;
    memset(ct, 0, sizeof(ct));
    r = fourStreams ? HUF_compress4X_repeat(dst, dstCap, src, n, 255, 11, wksp, sizeof(wksp), ct, &rep, flags)
                    : HUF_compress1X_repeat(dst, dstCap, src, n, 255, 11, wksp, sizeof(wksp), ct, &rep, flags);
    return HUF_isError(r) ? (size_t)-1 : r;
}

/* FSE_normalizeCount (fse_compress.c:465) */
size_t zref_fse_normalize(short* norm, unsigned tableLog, const unsigned* count, size_t total,
                          unsigned maxSym, unsigned useLowProb)
{
    size_t const r = FSE_normalizeCount(norm, tableLog, count, total, maxSym, useLowProb);
    return FSE_isError(r) ? (size_t)-1 : r;
}
unsigned zref_fse_optimal_tablelog(unsigned maxLog, size_t n, unsigned maxSym)
{
    return FSE_optimalTableLog(maxLog, n, maxSym);
}
size_t zref_fse_write_ncount(void* dst, size_t cap, const short* norm, unsigned maxSym, unsigned tableLog)
{
    size_t const r = FSE_writeNCount(dst, cap, norm, maxSym, tableLog);
    return FSE_isError(r) ? (size_t)-1 : r;
}

/* HUF_buildCTable_wksp + HUF_writeCTable_wksp: the tree description the literals section starts with (huf_compress.c:248-289) */
size_t zref_huf_write_table(void* dst, size_t cap, const unsigned* count, unsigned maxSym, unsigned maxNbBits, unsigned* logOut)
{
    HUF_CREATE_STATIC_CTABLE(ct, 255);
    static unsigned wksp[HUF_WORKSPACE_SIZE_U64 * 2];
    size_t const log = HUF_buildCTable_wksp(ct, count, maxSym, maxNbBits, wksp, sizeof(wksp));
    size_t r;
    if (HUF_isError(log)) return (size_t)-1;
    *logOut = (unsigned)log;
    r = HUF_writeCTable_wksp(dst, cap, ct, maxSym, (unsigned)log, wksp, sizeof(wksp));
    return HUF_isError(r) ? (size_t)-1 : r;
}
/* FSE_buildCTable_wksp (fse_compress.c:68-214): the state table and the per-symbol transforms of the encoding table */
size_t zref_fse_build_ctable(unsigned short* stateOut, int* dFind, unsigned* dBits, const short* norm, unsigned maxSym, unsigned tableLog)
{
    static unsigned ct[FSE_CTABLE_SIZE_U32(12, 255)];
    static unsigned wksp[FSE_BUILD_CTABLE_WORKSPACE_SIZE_U32(255, 12)];
    unsigned const tableSize = 1u << tableLog; unsigned s;
    size_t const r = FSE_buildCTable_wksp(ct, norm, maxSym, tableLog, wksp, sizeof(wksp));
    if (FSE_isError(r)) return (size_t)-1;
    memcpy(stateOut, ((const unsigned short*)ct) + 2, tableSize * 2);
    {   const FSE_symbolCompressionTransform* const tt = (const FSE_symbolCompressionTransform*)(ct + 1 + (tableLog ? tableSize >> 1 : 1));
        for (s = 0; s <= maxSym; s++) { dFind[s] = tt[s].deltaFindState; dBits[s] = tt[s].deltaNbBits; }
    }
    return 0;
}

unsigned zref_version(void) { return ZSTD_versionNumber(); }
