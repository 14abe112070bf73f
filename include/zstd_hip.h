/* zstd_hip.h — C ABI of libzstd_hip.so: the MI355X (gfx950) Zstandard block-compression core and, on the other side of the
 * same path, the batch frame decoder (section "decompression" below).
 *
 * This is the drop-in boundary for the hot path only.  Everything is plain C: opaque handle,
 * pointers and sizes; no C++/torch types.  Each entry point names the reference interface it stands in for
 * (paths relative to facebook/zstd).
 *
 * Unit of work: a *unit* = one independently compressed chunk of <= 128 KB (ZSTD_BLOCKSIZE_MAX) that becomes one
 * complete zstd frame holding one block — byte-identical to what the reference's
 *      ZSTD_compress2(cctx(level), dst, cap, chunk, chunkSize)          lib/zstd.h:603
 * emits for that chunk, i.e. to `zstd -b<level> -B<unitSize>` (programs/benchzstd.c:336-345).  Concatenated
 * frames are a valid .zst stream (RFC 8878 §3.1; lib/decompress/zstd_decompress.c:1068 iterates frames).
 *
 * Inputs above 128 KB have two more shapes, both byte-identical to the reference and both one standard frame per input (strategies
 * ZSTD_fast / ZSTD_dfast): zhip_compress_frames = ZSTD_compress2 of the whole input (a serial block chain, one workgroup per frame),
 * and zhip_compress_frames_mt = ZSTD_compress2 with ZSTD_c_nbWorkers >= 1 (the reference's job pool: independent jobs with an overlap
 * prefix, one workgroup per job — the shape in which ONE large input fills the GPU).
 *
 * Error convention = zstd's: size_t results are either a size or (size_t)-ZSTD_ErrorCode
 * (lib/common/error_private.h:55-60); test with zhip_isError().
 */
#ifndef ZSTD_HIP_H
#define ZSTD_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZHIP_UNIT_SIZE_MAX 131072          /* lib/zstd.h:143 ZSTD_BLOCKSIZE_MAX */
#define ZHIP_SEQUENCE_PRODUCER_ERROR ((size_t)(-1))   /* lib/zstd.h:2836 ZSTD_SEQUENCE_PRODUCER_ERROR */

typedef struct zhip_ctx_s zhip_ctx;

/* same layout as ZSTD_Sequence, lib/zstd.h:1287-1322 */
typedef struct { unsigned int offset, litLength, matchLength, rep; } zhip_Sequence;

/* ---- lifetime (no reference counterpart: owns the HIP device state that ZSTD_CCtx owns on the CPU side,
 *      lib/compress/zstd_compress.c:97 ZSTD_createCCtx) */
int          zhip_device_count(void);
zhip_ctx*    zhip_create(int device, size_t maxUnits);     /* scratch for up to maxUnits units per call */
void         zhip_destroy(zhip_ctx* ctx);
const char*  zhip_last_error(const zhip_ctx* ctx);

/* ---- error helpers = ZSTD_isError / ZSTD_getErrorName / ZSTD_compressBound (lib/zstd.h:236-243) */
unsigned     zhip_isError(size_t code);
const char*  zhip_getErrorName(size_t code);
size_t       zhip_compressBound(size_t srcSize, size_t unitSize);   /* sum of ZSTD_compressBound over the units */

/* ---- parameters = ZSTD_getCParams (lib/zstd.h:1877; lib/compress/zstd_compress.c:7150) for the supported rows
 * out[7] = windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy. returns 0, or -1 if the
 * level/size maps to a strategy this library does not implement (the binary-tree strategies btlazy2 .. btultra2, levels 13 and up;
 * fast, dfast, greedy, lazy and lazy2 are implemented). */
int          zhip_getCParams(int level, unsigned long long srcSize, unsigned out[7]);
/* the same with explicitly set parameters = what a CCtx holds after ZSTD_CCtx_setParameter(ZSTD_c_windowLog / chainLog / hashLog /
 * searchLog / minMatch / targetLength / strategy) (lib/compress/zstd_compress.c:710-768): cparams[7] in ZSTD_compressionParameters
 * order, 0 = "use the level's"; they replace the level's row BEFORE the source-size adjustment, as ZSTD_getCParamsFromCCtxParams
 * does (:1617-1644).  0 ok, 1 strategy not implemented on the device, 2 a value outside ZSTD_cParam_getBounds */
int          zhip_getCParams_explicit(int level, unsigned long long srcSize, const unsigned cparams[7], unsigned out[7]);

/* ---- frame-level drop-in for host buffers = ZSTD_compress2 per unit (lib/zstd.h:603), batch form.
 * src is cut into units of unitSize (last one ragged); dst receives the frames back to back.
 * unitSizes (optional, host, >= number of units) receives each frame's size. */
size_t       zhip_compress(zhip_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                           int level, size_t unitSize, size_t* unitSizes);

/* ---- same with device-resident buffers (HIP device pointers; hipStream_t passed as void*, NULL = ctx stream).
 * dstDev receives the packed frames; the total size is returned after the stream has been synchronised.
 * unitSizesDev (optional, device, uint32 per unit) receives frame sizes. */
size_t       zhip_compress_device(zhip_ctx* ctx, void* dstDev, size_t dstCapacity, const void* srcDev, size_t srcSize,
                                  int level, size_t unitSize, uint32_t* unitSizesDev, void* stream);

/* ---- both with explicit compression parameters (see zhip_getCParams_explicit): = ZSTD_compress2 on a CCtx whose advanced
 * parameters were set (lib/compress/zstd_compress.c:710-768).  cparams may be NULL (= the plain calls above).  Not implemented on
 * the device -> parameter_unsupported: strategies above lazy2, a windowLog smaller than the unit it is applied to (ZSTD_fast with
 * hashLog > 15 — beyond the unit kernel's LDS table — runs through the frame kernel's 24-bit LDS / HBM table, same bytes); values outside ZSTD_cParam_getBounds -> parameter_outOfBound. */
size_t       zhip_compress_params(zhip_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                  int level, const unsigned cparams[7], size_t unitSize, size_t* unitSizes);
size_t       zhip_compress_params_device(zhip_ctx* ctx, void* dstDev, size_t dstCapacity, const void* srcDev, size_t srcSize,
                                         int level, const unsigned cparams[7], size_t unitSize, uint32_t* unitSizesDev, void* stream);

/* ---- multi-block frames: input i = src[srcOffsets[i], srcOffsets[i+1]) becomes ONE standard frame holding
 * the blocks ZSTD_compress2 / ZSTD_compress emit for it on a fresh CCtx (lib/compress/zstd_compress.c:4520-4640 ZSTD_compress_frameChunk:
 * 128 KB blocks, 92 KB once the frame has saved 3 bytes; table, window, repcodes and Huffman table carried from block to block) —
 * byte-identical to the reference's single frame.  The block chain of a frame is serial: a frame is one workgroup, so the batch,
 * not the frame, is what fills the GPU (the per-unit calls above are the throughput path).  Implemented for the strategies ZSTD_fast
 * ... ZSTD_lazy2 (levels -N .. 12 of every size class; greedy / lazy / lazy2 with the row-hash matcher or the hash chain as zhip_set_row_matcher says:
 * zhip_frame_lazy.h carries chain / rows, nextToUpdate, the FSE tables repeated by cost and lazy2's block splitter across the blocks; a window — input, or job + overlap — below 1 GiB), inputs below 2 GiB each,
 * nFrames <= the context's maxUnits.  dstCapacity >= zhip_frames_bound().  frameSizes (optional) receives each frame's size. */
size_t       zhip_frames_bound(const unsigned long long* srcOffsets /* nFrames + 1 */, size_t nFrames);
size_t       zhip_compress_frames(zhip_ctx* ctx, void* dst, size_t dstCapacity, const void* src, const unsigned long long* srcOffsets,
                                  size_t nFrames, int level, const unsigned cparams[7] /* or NULL */, size_t* frameSizes);
size_t       zhip_compress_frames_device(zhip_ctx* ctx, void* dstDev, size_t dstCapacity, const void* srcDev, const unsigned long long* srcOffsets,
                                         size_t nFrames, int level, const unsigned cparams[7], uint32_t* frameSizesDev, void* stream);

/* ---- the same inputs as the frames ZSTD_compress2 emits with ZSTD_c_nbWorkers >= 1 (lib/compress/zstdmt_compress.c; the bytes do not
 * depend on the worker count): an input above 512 KB is cut into jobs of jobSize (0 = the reference's default 1 << max(20, windowLog+2);
 * ZSTD_c_jobSize semantics: clamped to 512 KB .. 1 GiB, raised to the overlap), each job compressed with the last
 * 1 << (windowLog - (9 - overlapLog)) bytes of its predecessor as prefix (overlapLog 0 = the strategy's default, 6 for fast/dfast;
 * ZSTD_c_overlapLog 1..9) and its own tables; the concatenation is one standard frame.  Jobs are independent, so ONE large input
 * fills the GPU: a workgroup per job.  Inputs of at most 512 KB come out as the plain frame above, as in the reference.  The context
 * needs maxUnits >= the total number of jobs.  Same strategies and limits as zhip_compress_frames. */
size_t       zhip_compress_frames_mt(zhip_ctx* ctx, void* dst, size_t dstCapacity, const void* src, const unsigned long long* srcOffsets,
                                     size_t nFrames, int level, const unsigned cparams[7] /* or NULL */, size_t jobSize, int overlapLog, size_t* frameSizes);
size_t       zhip_compress_frames_mt_device(zhip_ctx* ctx, void* dstDev, size_t dstCapacity, const void* srcDev, const unsigned long long* srcOffsets,
                                            size_t nFrames, int level, const unsigned cparams[7], size_t jobSize, int overlapLog,
                                            uint32_t* frameSizesDev, void* stream);

/* ---- host buffers over SEVERAL devices in one process: independent units shard across the GPUs of a node, one
 * kernel stream + copy stream + pinned staging per lane ($ZHIP_MULTI_LANES lanes per device, default 2: one lane's copies overlap the other's kernels), no collective;
 * finished chunks are gathered on the host in source order (destination offset = exclusive prefix sum of the sizes before).
 * devices[] may name the same device more than once (more lanes on it).  chunkUnits = units per chunk (0 = 1024 = 128 MB, with quarter chunks at both ends of a call).
 * The stream written to dst is byte-identical to zhip_compress's.  cparams may be NULL. */
typedef struct zhip_multi_s zhip_multi;
zhip_multi*  zhip_multi_create(const int* devices, int nDevices, size_t chunkUnits);
void         zhip_multi_destroy(zhip_multi* m);
int          zhip_multi_set_frame_checksum(zhip_multi* m, int enable);
int          zhip_multi_set_row_matcher(zhip_multi* m, int mode);      /* zhip_set_row_matcher on every lane and on the wide context (modes as there) */
size_t       zhip_compress_multi(zhip_multi* m, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                 int level, const unsigned cparams[7], size_t unitSize, size_t* unitSizes);
/* ONE input as the job-pool frame of zhip_compress_frames_mt (ZSTD_c_nbWorkers >= 1 semantics), its jobs spread over the lanes and
 * devices of m: copies of one chunk of jobs overlap the kernels of another, and several GPUs share one frame.  One device only, frame
 * checksum on, inputs up to 512 KB, or jobs larger than a lane's staging buffer: a single context's zhip_compress_frames_mt. */
size_t       zhip_compress_frame_mt_multi(zhip_multi* m, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                          int level, const unsigned cparams[7] /* or NULL */, size_t jobSize, int overlapLog);
const char*  zhip_multi_last_error(const zhip_multi* m);
double       zhip_multi_last_seconds(const zhip_multi* m);      /* wall time of the most recent zhip_compress_multi call */
/* where the most recent zhip_compress_multi call spent its time, seconds summed over all chunks and lanes (lanes overlap: the sum exceeds the
 * wall time): [0] host copy into pinned staging, [1] H2D, [2] kernels, [3] D2H (HIP events on the lane's stream), [4] waiting for earlier
 * chunks' sizes (the ordered gather), [5] host copy of the frames to their final place; [6] = number of chunks */
void         zhip_multi_last_stages(const zhip_multi* m, double out[7]);

/* ---- block-level plugin (B1) = ZSTD_sequenceProducer_F, lib/zstd.h:2838; contrib/externalSequenceProducer.
 * zhip_sequence_producer has exactly that signature; pass the zhip_ctx as sequenceProducerState:
 *      ZSTD_registerSequenceProducer(cctx, zhip_ctx, zhip_sequence_producer);        lib/zstd.h:2866-2871
 * One kernel launch per callback is latency-bound, so zhip_prepare_sequences() parses every block of a buffer in
 * one launch beforehand; the callback then serves blocks of that buffer from the cache and launches only for
 * blocks it has not seen.  Failures map to ZHIP_SEQUENCE_PRODUCER_ERROR so that
 * ZSTD_c_enableSeqProducerFallback works (lib/compress/zstd_compress.c:3338-3356). */
size_t       zhip_sequence_producer(void* sequenceProducerState, zhip_Sequence* outSeqs, size_t outSeqsCapacity,
                                    const void* src, size_t srcSize, const void* dict, size_t dictSize,
                                    int compressionLevel, size_t windowSize);
size_t       zhip_prepare_sequences(zhip_ctx* ctx, const void* src, size_t srcSize, size_t blockSize, int level);

/* Stage-1 only, device-resident: parse every unit, keep the result in ctx; fetch one unit's sequences in the
 * ZSTD_generateSequences format (lib/zstd.h:1593-1597; block delimiter {0,lastLits,0,0} appended). */
size_t       zhip_parse_device(zhip_ctx* ctx, const void* srcDev, size_t srcSize, int level, size_t unitSize, void* stream);
size_t       zhip_get_sequences(zhip_ctx* ctx, size_t unitIndex, zhip_Sequence* out, size_t capacity);

/* ---- ZSTD_c_checksumFlag (lib/zstd.h:444; what the zstd CLI turns on by default, programs/fileio.c:287): when enabled every
 * frame this context emits carries the 32-bit content checksum (low half of XXH64, computed on the device: k_xxh64) and the
 * descriptor bit, exactly as lib/compress/zstd_compress.c:4637 / :5297-5303 write them.  Sticky until changed.  Returns 0. */
int          zhip_set_frame_checksum(zhip_ctx* ctx, int enable);
/* = ZSTD_c_useRowMatchFinder (lib/zstd.h, experimental): which match finder the strategies greedy / lazy / lazy2 use.
 * 0 (default, ZSTD_ps_auto): the reference's default — the row-hash matcher when windowLog > 14 (lib/compress/zstd_compress.c:237-253),
 * with the hash salt of a FRESH CCtx (bytes = ZSTD_compress2 on a fresh CCtx per unit; a reused reference CCtx mixes the previous
 * frames' hashes into its salt, :1964-1975); 1 (ZSTD_ps_enable): the same, except that a unit the reference gives windowLog <= 14 is
 * REFUSED (parameter_unsupported) — the reference would run its row matcher there too (:244), the device has none for such windows;
 * 2 (ZSTD_ps_disable): the hash-chain matcher; -1: back to the mode the context was created with.
 * The environment variable ZHIP_ROW_MATCHER=disable sets 2 as a context's initial mode.  Returns 0, or 1 for another value. */
int          zhip_set_row_matcher(zhip_ctx* ctx, int mode);
/* The row matcher's two-pass prediction: a first parse marks the positions the 384-position rule and lazy skipping will leave
 * un-inserted, the per-position records are recomputed without them, and the exact parse redoes a search live only where prediction and truth differ.
 * Same bytes with it on or off (this call sets it).  Units: off by default — a live search there reads the row's list of positions
 * and one parse is the fastest form.  Frames: on by default, behind a probe — a window whose first 32 KB leave nothing un-inserted is parsed once.
 * units / frames: 1 on, 0 off, -1 unchanged.  returns 0, or 1 for a bad value. */
int          zhip_set_prediction(zhip_ctx* ctx, int units, int frames);
/* the FRAME kernels' live rows (the row matcher's rows as the reference keeps them): 1 on (default), 0 = their live searches walk the
 * links (what a context without the rows' arena does).  Same bytes either way; the unit kernels keep no rows.  returns 0. */
int          zhip_set_live_rows(zhip_ctx* ctx, int on);

/* ---- seekable container (contrib/seekable_format/zstd_seekable_compression_format.md): independent frames followed by a
 * skippable frame holding the seek table — the natural on-disk form of frame-per-unit output; the reference's
 * ZSTD_seekable_decompress (contrib/seekable_format/zstdseek_decompress.c) reads it.  zhip_write_seek_table emits exactly
 * the bytes ZSTD_seekable_writeSeekTable (zstdseek_compress.c) does for the same frame log. */
size_t       zhip_seek_table_bound(size_t nFrames, int withChecksum);
size_t       zhip_write_seek_table(void* dst, size_t dstCapacity, const unsigned* compressedSizes, const unsigned* decompressedSizes,
                                   const unsigned* checksums /* NULL = no checksum column */, size_t nFrames);
/* zhip_compress (host buffers) + the seek table appended; per-frame checksums go into the table when
 * zhip_set_frame_checksum is on.  dstCapacity >= zhip_compressBound(..) + zhip_seek_table_bound(nUnits, 1). */
size_t       zhip_compress_seekable(zhip_ctx* ctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, size_t unitSize);

/* ---- dictionary compression of many small records (BASELINE configs[4]): what
 *      cdict = ZSTD_createCDict(dict, dictSize, level);                      lib/zstd.h:979
 *      ZSTD_CCtx_refCDict(cctx, cdict); ZSTD_compress2(cctx, ..record..)     lib/zstd.h:1102, :603    per record
 * produces — one frame per record, byte-identical.  Records up to the reference's attach cut-off (8 KB for strategy fast,
 * 16 KB for dfast: lib/compress/zstd_compress.c:2289-2315) take its ATTACH mode (zstd_fast.c:483-678,
 * zstd_double_fast.c:328-547); larger ones, up to 128 KB, its COPY mode: private copies of the CDict's tables and the
 * extDict block compressors (zstd_compress.c:2395-2470, zstd_fast.c:709-960, zstd_double_fast.c:551-759), one source per GPU
 * lane.  The CDict's tables are built once on the host exactly like ZSTD_createCDict builds them (zstd_fast.c:16-49,
 * zstd_double_fast.c:18-54) and uploaded.
 * Raw-content and ZDICT-format dictionaries (entropy tables, repcodes, dictID: zstd_compress.c:4986-5118); levels whose
 * CDict row is strategy fast or dfast (levels -N..4).  Lazy-strategy dictionaries return NULL, records above 128 KB
 * parameter_unsupported — there is no CPU fallback. */
typedef struct zhip_cdict_s zhip_cdict;
zhip_cdict*  zhip_create_cdict(int device, const void* dict, size_t dictSize, int level);
void         zhip_free_cdict(zhip_cdict* cdict);
/* a context whose scratch is sized for `maxRecords` records of `maxTotalBytes` source bytes in one call */
zhip_ctx*    zhip_create_for_records(int device, size_t maxRecords, size_t maxTotalBytes);
/* sum of ZSTD_compressBound over the records; recOffsets has nRec+1 entries (record i = [recOffsets[i], recOffsets[i+1])) */
size_t       zhip_records_bound(const unsigned long long* recOffsets, size_t nRec);
/* records and destination resident in HBM; recOffsets is a HOST array; frameSizesDev (optional) gets nRec u32 sizes;
 * frames are packed back to back into dstDev.  Returns the total compressed size. */
size_t       zhip_compress_records_device(zhip_ctx* ctx, const zhip_cdict* cdict, void* dstDev, size_t dstCapacity,
                                          const void* srcDev, const unsigned long long* recOffsets, size_t nRec,
                                          uint32_t* frameSizesDev, void* stream);
/* host buffers (staged over PCIe); frameSizes (optional) gets nRec sizes */
size_t       zhip_compress_records(zhip_ctx* ctx, const zhip_cdict* cdict, void* dst, size_t dstCapacity,
                                   const void* src, const unsigned long long* recOffsets, size_t nRec, size_t* frameSizes);

/* ---- decompression: the step on the other side of the path, batch form of
 *      ZSTD_decompress (lib/zstd.h:168) / ZSTD_decompressDCtx (:294) / ZSTD_decompress_usingDDict (:1013).
 * Any RFC 8878 frame is accepted (what this library emits and what the reference emits: several blocks per frame, repeat /
 * treeless modes, checksums, dictionaries); frames are independent work items, one workgroup each.  Per-frame failures
 * carry zstd's error codes (corruption_detected 20, checksum_wrong 22, dictionary_wrong 32, dstSize_tooSmall 70, ...). */
typedef struct zhip_dctx_s  zhip_dctx;                      /* device state the way ZSTD_DCtx owns it, lib/zstd.h:285 */
typedef struct zhip_ddict_s zhip_ddict;                     /* = ZSTD_DDict, lib/zstd.h:998: content + entropy tables in decoding form */
zhip_dctx*   zhip_create_dctx(int device);
void         zhip_free_dctx(zhip_dctx* dctx);
const char*  zhip_dctx_last_error(const zhip_dctx* dctx);
zhip_ddict*  zhip_create_ddict(int device, const void* dict, size_t dictSize);   /* raw-content or ZDICT format (zstd_decompress.c:1476-1500) */
void         zhip_free_ddict(zhip_ddict* ddict);
unsigned     zhip_ddict_id(const zhip_ddict* ddict);        /* = ZSTD_getDictID_fromDDict, lib/zstd.h:1039 */
/* frame walker over a HOST buffer of concatenated frames = ZSTD_findFrameCompressedSize (lib/zstd.h:214) +
 * ZSTD_getFrameContentSize (:215) + ZSTD_decompressBound's per-frame term (:1520); skippable frames are stepped over.
 * Arrays are optional, filled for the first maxFrames frames; contentSizes[i] = ~0ull when the header does not state it.
 * returns the number of frames, or an error (srcSize_wrong when the buffer does not end on a frame boundary). */
size_t       zhip_find_frames(const void* src, size_t srcSize, unsigned long long* srcOffsets, unsigned long long* srcSizes,
                              unsigned long long* contentSizes, unsigned long long* contentBounds, size_t maxFrames);
size_t       zhip_frame_compressed_size(const void* src, size_t srcSize);   /* = ZSTD_findFrameCompressedSize: the first frame of src */
/* device-resident buffers; the four descriptor arrays are HOST arrays of nFrames entries: frame i occupies
 * srcDev[srcOffsets[i] .. +srcSizes[i]) and decodes to dstDev[dstOffsets[i] .. +dstCapacities[i]).  statusOut / sizesOut
 * (optional, host) receive each frame's zstd error code (0 = ok) and decoded size.  Returns the total decoded size after
 * the stream has been synchronised, or the first frame's error. */
size_t       zhip_decompress_frames_device(zhip_dctx* dctx, const zhip_ddict* ddict, void* dstDev, const unsigned long long* dstOffsets,
                                           const unsigned long long* dstCapacities, const void* srcDev, const unsigned long long* srcOffsets,
                                           const unsigned long long* srcSizes, size_t nFrames, unsigned* statusOut,
                                           unsigned long long* sizesOut, void* stream);
/* host buffers = ZSTD_decompress(dst, cap, src, srcSize): every frame of src, contents back to back (ddict may be NULL) */
size_t       zhip_decompress(zhip_dctx* dctx, const zhip_ddict* ddict, void* dst, size_t dstCapacity, const void* src, size_t srcSize);
/* = ZSTD_seekable_decompress (contrib/seekable_format/zstd_seekable.h:166): len bytes of the decompressed data at `offset` from
 * a seekable file (frames + seek table, what zhip_compress_seekable or the reference's seekable compressor writes) in a HOST
 * buffer; only the frames overlapping the request are decoded (one batch), the table's checksums are verified when present. */
size_t       zhip_seekable_read(zhip_dctx* dctx, void* dst, size_t len, const void* src, size_t srcSize, unsigned long long offset);
/* HIP-event durations (ms) of the most recent call: t[0] = k_decode, t[1] = checksum verification (0 when no frame has one) */
void         zhip_dctx_last_timing(const zhip_dctx* dctx, double t[2]);
/* One LARGE frame is decoded block-parallel (zstd_amd/csrc/zhip_decode_big.h: symbolic repeat offsets + a scan, pointer jumping over a
 * copy map) instead of by one workgroup: frames without a dictionary that hold at least minContent bytes of content (stated in the header, or bounded by the destination slot)
 * (default 8 MiB; 0 = never).  Same bytes, same errors: whatever that path declines goes through the per-frame decoder.
 * zhip_dctx_last_bigframe: [0] frames of the last call decoded block-parallel, [1] frames that fell back, [2] pointer-jumping rounds, [3] blocks. */
void         zhip_dctx_set_bigframe_min(zhip_dctx* dctx, unsigned long long minContent);
void         zhip_dctx_last_bigframe(const zhip_dctx* dctx, unsigned out[4]);

/* ---- measurement: HIP-event durations (ms) of the kernels of the most recent call on this ctx
 * t[0] = match finder, t[1] = entropy + frame assembly, t[2] = output compaction, t[3] = whole device pipeline */
void         zhip_last_timing(const zhip_ctx* ctx, double t[4]);
/* hash-chain strategies (levels 5+): the match-finder stage of the most recent call split by kernel, ms summed over
 * its chunks: [0] k_hc_chain (links), [1] k_hc_search (best match per position), [2] k_parse_lazy (parser) */
void         zhip_last_hc_timing(zhip_ctx* ctx, double t[3]);

/* stats[0] units, [1] source bytes, [2] compressed bytes, [3] sequences, [4] literal bytes of the most recent call */
size_t       zhip_last_stats(zhip_ctx* ctx, unsigned long long stats[5]);

/* ---- stage-test hooks (tests/test_gpu_tables.py): the device's wave-wide entropy-table builders run on caller-supplied
 * histograms, one 64-thread workgroup per case, so that each stage can be compared with the reference's own stage function —
 * HUF_buildCTable_wksp + HUF_writeCTable_wksp (lib/compress/huf_compress.c:756, :248) and FSE_normalizeCount + FSE_writeNCount +
 * FSE_buildCTable_wksp (lib/compress/fse_compress.c:465, :329, :68).  Host buffers; 0 or an error code.
 *   huf: counts[nCases][256], maxSyms[nCases] -> codes[nCases][256] (value << 8 | nbBits), hdrs[nCases][136], meta[nCases][2] = {tableLog, header size}
 *   fse: counts[nCases][64], params[nCases][4] = {total, maxSym, tableLog, useLowProbCount} -> norms[nCases][64], ncounts[nCases][64],
 *        meta[nCases][2] = {normalize result 1 / 0 (single symbol) / -1, NCount size}, tables[nCases] of tableStride bytes each:
 *        {u16 state[512]; i32 deltaFindState[56]; u32 deltaNbBits[56]; u32 tableLog} */
size_t       zhip_test_huf_tables(zhip_ctx* ctx, const unsigned* counts, const unsigned* maxSyms, unsigned nCases, unsigned maxNbBits,
                                  unsigned* codes, unsigned char* hdrs, unsigned* meta);
size_t       zhip_test_fse_tables(zhip_ctx* ctx, const unsigned* counts, const unsigned* params, unsigned nCases,
                                  short* norms, unsigned char* ncounts, int* meta, void* tables, size_t tableStride);

#ifdef __cplusplus
}
#endif
#endif
