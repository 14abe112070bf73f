/* zstd_hip_dropin.h — the ZSTD_* entry points libzstd_hipshim.so exports (frame-level boundary B2).
 *
 * The prototypes are the reference's own (facebook/zstd lib/zstd.h, line cited per function): a program written
 * against zstd.h compiles unchanged and links -lzstd_hipshim instead of -lzstd for the calls below.  Everything is
 * served by the gfx950 kernels of libzstd_hip.so; there is NO CPU path, so a parameter set the device core does not
 * implement returns the reference's own error code (ZSTD_error_parameter_unsupported, lib/zstd_errors.h:74) instead
 * of silently compressing on the host.
 *
 * Output contract
 *   srcSize <= 128 KB : ONE frame, byte-identical to the reference's ZSTD_compress2 with the same parameters set on its CCtx —
 *                       ZSTD_c_compressionLevel and the advanced parameters ZSTD_c_windowLog / chainLog / hashLog / searchLog /
 *                       minMatch / targetLength / strategy (applied as ZSTD_getCParamsFromCCtxParams applies them,
 *                       lib/compress/zstd_compress.c:1617-1644).
 *                       Strategies greedy / lazy / lazy2 (default at levels 5-12): both of the reference's match finders run on
 *                       the device — the row-hash matcher where the reference picks it by default (windowLog > 14,
 *                       zstd_compress.c:237-253) and the hash chain otherwise or with ZSTD_c_useRowMatchFinder = disable.
 *                       The row matcher's hash is SALTED per CCtx reset (zstd_compress.c:1964-1975): the device uses the salt
 *                       of a fresh CCtx, so the bytes equal ZSTD_compress() / ZSTD_compress2 on a NEW CCtx; a reference CCtx
 *                       that already compressed something else has another salt and may pick other (equally valid) matches.
 *                       Not implemented on the device -> parameter_unsupported: strategies above lazy2,
 *                       a windowLog smaller than the input, a dictionary whose CDict
 *                       row is a lazy strategy, dictionary + source above 128 KB.
 *   srcSize  > 128 KB : the source is cut into 128 KB units, each an independent frame (content size in every frame
 *                       header), emitted back to back.  That is a valid zstd stream (RFC 8878 3.1: frames may be
 *                       concatenated; ZSTD_decompress decodes it, lib/decompress/zstd_decompress.c:1068) and equals
 *                       `zstd -b<level> -B128K` chunking, but it is NOT the reference's single shared-window frame.
 *                       ZSTD_getFrameContentSize() of the stream reports the first unit only; use
 *                       ZSTD_findDecompressedSize() (lib/zstd.h:1458) for the total.
 *                       With ZHIP_c_singleFrame = 1 (or $ZHIP_SINGLE_FRAME=1) and any strategy up to ZSTD_lazy2 (levels -N .. 12)
 *                       the output IS the reference's single frame, byte for byte: one frame header,
 *                       128 KB / 92 KB blocks (ZSTD_lazy2: the fingerprint splitter's borders) sharing the window, the match
 *                       finder's table (hash table, hash chain or rows), the repcodes, the Huffman table and — greedy and above —
 *                       the FSE tables (zstd_compress.c:4520-4640; zhip_compress_frames).  The block chain of one frame is
 *                       serial — one workgroup — so this is the fidelity mode, not the throughput mode; strategies above
 *                       lazy2 are parameter_unsupported.
 *                       With ZSTD_c_nbWorkers >= 1 (and ZSTD_c_jobSize / ZSTD_c_overlapLog) and a source above 512 KB the output is
 *                       the reference's multi-threaded frame, byte for byte (lib/compress/zstdmt_compress.c: jobs of the job size,
 *                       each with the overlap as prefix; the bytes do not depend on the worker count) — jobs are independent, a
 *                       workgroup each, so this mode is both the reference's bytes and parallel inside one frame
 *                       (zhip_compress_frames_mt).  Same strategies as ZHIP_c_singleFrame.
 */
#ifndef ZSTD_HIP_DROPIN_H
#define ZSTD_HIP_DROPIN_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ZSTD_CCtx_s ZSTD_CCtx;                                   /* lib/zstd.h:262 */
typedef struct ZSTD_CDict_s ZSTD_CDict;                                 /* lib/zstd.h:965 */
typedef enum { ZSTD_reset_session_only = 1, ZSTD_reset_parameters = 2, ZSTD_reset_session_and_parameters = 3 } ZSTD_ResetDirective;   /* :569-573 */
/* ZSTD_cParameter values this shim understands (lib/zstd.h:331-522); all others -> parameter_unsupported */
enum { ZSTD_c_compressionLevel = 100, ZSTD_c_windowLog = 101, ZSTD_c_hashLog = 102, ZSTD_c_chainLog = 103, ZSTD_c_searchLog = 104,
       ZSTD_c_minMatch = 105, ZSTD_c_targetLength = 106, ZSTD_c_strategy = 107,
       ZSTD_c_contentSizeFlag = 200, ZSTD_c_checksumFlag = 201, ZSTD_c_dictIDFlag = 202, ZSTD_c_nbWorkers = 400, ZSTD_c_jobSize = 401, ZSTD_c_overlapLog = 402,
       ZSTD_c_useRowMatchFinder = 1011 /* = ZSTD_c_experimentalParam14: 0 auto, 1 enable, 2 disable */,
       ZHIP_c_singleFrame = 100001 /* not a reference parameter: 1 = sources above 128 KB as one multi-block frame (see above) */ };

ZSTD_CCtx*  ZSTD_createCCtx(void);                                                                 /* lib/zstd.h:263 */
size_t      ZSTD_freeCCtx(ZSTD_CCtx* cctx);                                                        /* :264 */
size_t      ZSTD_CCtx_setParameter(ZSTD_CCtx* cctx, int param, int value);                         /* :550 */
size_t      ZSTD_CCtx_reset(ZSTD_CCtx* cctx, ZSTD_ResetDirective reset);                           /* :589 */
size_t      ZSTD_compress2(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize);   /* :603 */
size_t      ZSTD_compressCCtx(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);   /* :274 */
size_t      ZSTD_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);   /* :155 */
size_t      ZSTD_compressBound(size_t srcSize);                                                    /* :236 (here: the bound of the frame-per-unit stream, >= the reference's) */
unsigned    ZSTD_isError(size_t code);                                                             /* :242 */
const char* ZSTD_getErrorName(size_t code);                                                        /* :243 */
/* dictionaries: raw-content and ZDICT-format, CDict levels whose row is fast/dfast, sources up to 128 KB (attach mode below
 * the reference's cut-off of 8 KB fast / 16 KB dfast, copy mode above it) — byte-identical to the reference; anything else ->
 * NULL / parameter_unsupported */
ZSTD_CDict* ZSTD_createCDict(const void* dictBuffer, size_t dictSize, int compressionLevel);       /* :979 */
size_t      ZSTD_freeCDict(ZSTD_CDict* CDict);                                                     /* :985 */
size_t      ZSTD_CCtx_refCDict(ZSTD_CCtx* cctx, const ZSTD_CDict* cdict);                          /* :1102 */
/* streaming entry point, one-shot form only (see zstd_shim.c): first call of the frame with ZSTD_e_end, all input present,
 * output room >= ZSTD_compressBound(input) -> the bytes of ZSTD_compress2, returns 0; everything else -> parameter_unsupported */
typedef struct ZSTD_inBuffer_s  { const void* src; size_t size; size_t pos; } ZSTD_inBuffer;       /* lib/zstd.h:681 */
typedef struct ZSTD_outBuffer_s { void* dst; size_t size; size_t pos; } ZSTD_outBuffer;            /* :687 */
typedef enum { ZSTD_e_continue = 0, ZSTD_e_flush = 1, ZSTD_e_end = 2 } ZSTD_EndDirective;          /* :763-774 */
typedef ZSTD_CCtx ZSTD_CStream;                                                                    /* :756 */
size_t        ZSTD_compressStream2(ZSTD_CCtx* cctx, ZSTD_outBuffer* output, ZSTD_inBuffer* input, ZSTD_EndDirective endOp);   /* :803 */
ZSTD_CStream* ZSTD_createCStream(void);                                                            /* :759 */
size_t        ZSTD_freeCStream(ZSTD_CStream* zcs);                                                 /* :760 */
size_t        ZSTD_initCStream(ZSTD_CStream* zcs, int compressionLevel);                           /* :842 */
size_t        ZSTD_CStreamInSize(void);                                                            /* :822 */
size_t        ZSTD_CStreamOutSize(void);                                                           /* :823 */
size_t      ZSTD_compress_usingCDict(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, const ZSTD_CDict* cdict);   /* :992 */
/* decompression (served by k_decode; any RFC 8878 frame — this library's and the reference's) */
typedef struct ZSTD_DCtx_s ZSTD_DCtx;                                                              /* :285 */
typedef struct ZSTD_DDict_s ZSTD_DDict;                                                            /* :998 */
#define ZSTD_CONTENTSIZE_UNKNOWN 0xFFFFFFFFFFFFFFFFULL                                                       /* :194 */
#define ZSTD_CONTENTSIZE_ERROR   0xFFFFFFFFFFFFFFFEULL                                                       /* :195 */
size_t      ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize);   /* :168 every frame of src */
ZSTD_DCtx*  ZSTD_createDCtx(void);                                                                 /* :286 */
size_t      ZSTD_freeDCtx(ZSTD_DCtx* dctx);                                                        /* :287 */
size_t      ZSTD_decompressDCtx(ZSTD_DCtx* dctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize);   /* :294 */
ZSTD_DDict* ZSTD_createDDict(const void* dictBuffer, size_t dictSize);                             /* :1003 */
size_t      ZSTD_freeDDict(ZSTD_DDict* ddict);                                                     /* :1008 */
size_t      ZSTD_decompress_usingDDict(ZSTD_DCtx* dctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, const ZSTD_DDict* ddict);   /* :1013 */
unsigned    ZSTD_getDictID_fromDDict(const ZSTD_DDict* ddict);                                     /* :1039 */
unsigned long long ZSTD_getFrameContentSize(const void* src, size_t srcSize);                      /* :196 first frame only */
unsigned long long ZSTD_findDecompressedSize(const void* src, size_t srcSize);                     /* :1458 all frames */
size_t      ZSTD_findFrameCompressedSize(const void* src, size_t srcSize);                         /* :214 */
unsigned long long ZSTD_decompressBound(const void* src, size_t srcSize);                          /* :1473 upper bound of all frames' content */
unsigned    ZSTD_isFrame(const void* buffer, size_t size);                                         /* :2361 zstd or skippable frame magic */
unsigned    ZSTD_getDictID_fromFrame(const void* src, size_t srcSize);                             /* :1051 0 = not stated */
int         ZSTD_minCLevel(void);                                                                  /* :244 */
int         ZSTD_maxCLevel(void);                                                                  /* :245 (highest level the device core implements) */
int         ZSTD_defaultCLevel(void);                                                              /* :246 */

#ifdef __cplusplus
}
#endif
#endif
