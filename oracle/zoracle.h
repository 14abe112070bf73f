/* oracle/zoracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, single-threaded CPU restatement of the reference's level-1..4 block-compression core
 * (facebook/zstd @ /root/reference, v1.5.6+dev).  It exists so that tests, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg have a checker that travels to the GPU box as source.  It is pinned
 * byte-for-byte against the real reference (oracle/_ref, built from /root/reference) by
 * tests/test_oracle_vs_reference.py and by the committed fixtures in tests/golden/.
 *
 * The product (zstd_amd/, include/) must never include, link or call this.
 */
#ifndef ZORACLE_H
#define ZORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    unsigned windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy;
} zo_cparams;

/* one parsed sequence, full-width fields (the reference's U16 + longLength trick is folded in at coding time) */
typedef struct {
    uint32_t litLength;    /* literals preceding the match */
    uint32_t matchLength;  /* real match length (>= 3)      */
    uint32_t offBase;      /* 1..3 = repcode id, else offset + 3 (lib/compress/zstd_compress_internal.h:654-662) */
} zo_seq;

#define ZO_BLOCK_MAX 131072u
#define ZO_ERROR ((size_t)-1)

/* lib/compress/zstd_compress.c:7123 + :1466 (dictSize = 0, mode = noAttachDict), rows from clevels.h */
int    zo_get_cparams(int level, unsigned long long srcSize, zo_cparams* out);
/* lib/zstd.h:235 ZSTD_COMPRESSBOUND */
size_t zo_compress_bound(size_t n);

/* Stage 1.  Parse ONE block with no history (fresh tables, rep = {1,4,8}), strategy fast or dfast.
 * seqs[cap], lits[n] filled; returns nbSeq (ZO_ERROR on overflow); *litSize = all literals incl. trailing ones;
 * repOut = repcode history after the block.   zstd_fast.c:192-423, zstd_double_fast.c:105-323 */
size_t zo_parse_block(const zo_cparams* cp, const uint8_t* src, size_t n,
                      zo_seq* seqs, size_t cap, uint8_t* lits, size_t* litSize, uint32_t repOut[3]);

/* Same sequences expressed the way ZSTD_generateSequences reports them (zstd_compress.c:3371-3440):
 * out[4*i] = {offset, litLength, matchLength, rep}, final delimiter {0,lastLits,0,0}. returns count. */
size_t zo_sequences_public(const zo_cparams* cp, const uint8_t* src, size_t n, uint32_t* out, size_t capSeqs);

/* Stage 2 pieces (each usable on its own by the per-kernel parity tests). */
size_t zo_hist(unsigned count[256], unsigned* maxSym, const uint8_t* src, size_t n);             /* hist.c:29 */
/* code lengths for a histogram; returns max nbBits. huf_compress.c:756 */
unsigned zo_huf_build(const unsigned* count, unsigned maxSym, unsigned maxNbBits, uint8_t nbBits[256]);
/* literals section (header + payload). zstd_compress_literals.c:129 (no previous table) */
size_t zo_compress_literals(uint8_t* dst, size_t cap, const uint8_t* lits, size_t litSize,
                            const zo_cparams* cp, int suspectUncompressible);
/* sequences section (nbSeq header .. bitstream). zstd_compress.c:2934-2997 (no previous tables).
 * returns size, or 0 for the "old decoder" corner (:2987). */
size_t zo_compress_sequences(uint8_t* dst, size_t cap, const zo_seq* seqs, size_t nbSeq, const zo_cparams* cp);
int    zo_fse_normalize(short* norm, unsigned tableLog, const unsigned* count, size_t total,
                        unsigned maxSym, unsigned useLowProb);                                   /* fse_compress.c:465 */

/* Whole frame whose content is a single block (n <= 128 KB): what ZSTD_compress2 emits for an independent unit
 * at `level` with library defaults (no checksum, content size on).  zstd_compress.c:4527-4672, :5270 */
size_t zo_compress_unit(void* dst, size_t cap, const void* src, size_t n, int level);
size_t zo_compress_unit_params(void* dst, size_t cap, const void* src, size_t n, const zo_cparams* cp);
/* one frame per chunk, concatenated (= `zstd -b# -B<chunk>` byte stream). sizes[] optional. */
size_t zo_compress_chunks(int level, size_t chunkSize, const void* src, size_t n,
                          void* dst, size_t cap, size_t* sizes, size_t maxChunks);

/* frame checksum (ZSTD_c_checksumFlag): XXH64 restated; zo_frame_add_checksum turns a frame made by the functions above into
 * the one the reference emits with the flag set (needs 4 spare bytes after frameSize) */
uint64_t zo_xxh64(const void* src, size_t n, uint64_t seed);
size_t   zo_frame_add_checksum(void* frame, size_t frameSize, const void* src, size_t n);

/* Dictionary compression, attach mode (zstd_fast.c:483-678, zstd_double_fast.c:328-547, fill functions
 * zstd_fast.c:16-49 / zstd_double_fast.c:18-54, parameters zstd_compress.c:1466-1602 + :2318-2376): what
 * ZSTD_createCDict(dict, size, level) + ZSTD_CCtx_refCDict + ZSTD_compress2 emit for one small source. */
typedef struct zo_cdict_s zo_cdict;
zo_cdict* zo_cdict_create(const void* dict, size_t dictSize, int level);
void      zo_cdict_free(zo_cdict* cd);
int       zo_cdict_params(const zo_cdict* cd, size_t srcSize, zo_cparams* out);   /* -1: the reference would not attach */
size_t    zo_compress_unit_cdict(void* dst, size_t cap, const void* src, size_t n, const zo_cdict* cd);

/* programs/datagen.c:144 RDG_genBuffer and :155 RDG_genStdout restated (input generators for tests/bench) */
void zo_datagen(void* buf, size_t size, double matchProba, double litProba, unsigned seed);

#ifdef __cplusplus
}
#endif
/* strategies greedy / lazy / lazy2: 1 (default) = the reference's default matcher selection (row hash when windowLog > 14, with the
 * salt of a fresh CCtx), 0 = ZSTD_c_useRowMatchFinder = ZSTD_ps_disable (hash chain) */
void zo_set_row_matcher(int enable);
unsigned long long zo_fresh_hash_salt(void);

/* ONE frame for a source of any size (multi-block: shared window, hash table, repcodes, previous Huffman table; 92 KB blind split) =
 * ZSTD_compress2 on the whole source.  Strategy ZSTD_fast only (else ZO_ERROR).  cap >= zo_frame_bound(n). */
size_t zo_frame_bound(size_t n);
size_t zo_compress_frame(void* dst, size_t cap, const void* src, size_t n, int level);
size_t zo_compress_frame_params(void* dst, size_t cap, const void* src, size_t n, const zo_cparams* cp);
/* the frame ZSTD_compress2 emits with ZSTD_c_nbWorkers >= 1 (jobs sharing only an overlap prefix, zstdmt_compress.c); jobSize / overlapLog 0 = defaults */
size_t zo_compress_frame_cdict(void* dst, size_t cap, const void* src, size_t n, const zo_cdict* cd);   /* refCDict + compress2 above 128 KB (copy mode, fast / dfast) */
size_t zo_compress_frame_mt_params(void* dst, size_t cap, const void* src, size_t n, const zo_cparams* cp,
                                   unsigned long long jobSize, int overlapLog, int checksumFlag);
size_t zo_mt_job_size(const zo_cparams* cp, unsigned long long jobSize);
size_t zo_mt_overlap_size(const zo_cparams* cp, int overlapLog);

#endif
