/* oracle/ref_bench.c — TEST/BENCH INFRASTRUCTURE ONLY (bench.py's `cpu_baseline` leg, kind "reference").
 *
 * Times the REAL reference (oracle/_ref/libzstd_ref.so, built from /root/reference) on the host cores the
 * same way programs/benchzstd.c does for `zstd -b# -B128K` (benchzstd.c:336-345, :567, benchfn.c:107-256):
 * one ZSTD_compress2 call per chunk on a reused CCtx, repeated runs, fastest run kept, MB = 1e6 source bytes.
 *
 *   zref_bench bench  <level> <chunkSize> <totalBytes> <P%> <seed> <seconds> <threads>
 *        input = RDG_genBuffer(totalBytes, P/100, 0.0, seed) (programs/datagen.c:144); with threads>1 each
 *        thread compresses a disjoint contiguous shard of the chunks with its own CCtx.
 *        prints one JSON line.
 *   zref_bench file   <level> <chunkSize> <path> <seconds> <threads>   : same, input read from a file
 *   zref_bench cfile  <level> <chunkSize> <inPath> <outPath> <threads> : the file compressed ONCE, one frame per chunk, on
 *        `threads` cores, the frames written back to back to outPath (bench.py hashes that stream: full-size parity)
 *   zref_bench mtfile <level> <workers> <inPath> <outPath> <jobSize>  : the WHOLE file as ONE frame by ZSTD_compress2 with
 *        ZSTD_c_nbWorkers = workers (the reference's own job pool, lib/compress/zstdmt_compress.c; jobSize 0 = default), timed once
 *        and written to outPath (bench.py's job_pool_frame leg hashes it; the bytes do not depend on the worker count)
 *   zref_bench dfile  <level> <chunkSize> <path> <seconds> <threads>   : DECODE speed (`zstd -b#` second figure,
 *        benchzstd.c:380-420): the file is compressed once into one frame per chunk, then ZSTD_decompressDCtx per frame on a
 *        reused DCtx is timed; threads split the frames.
 *   zref_bench ddict  <level> <dictPath> <recordsPath> <offsetsPath> <seconds> <threads> : DECODE speed of the same record frames
 *        (ZSTD_createDDict + ZSTD_decompress_usingDDict per record)
 *   zref_bench ctile  <level> <chunkSize> <basePath> <copies> <shift> <totalBytes> <outPath> <threads> : like cfile, but the input is
 *        built here the way bench.py tiles a corpus on the device (copy c starts at offset c*shift mod len, wraps) — the 13 GiB of
 *        BASELINE configs[2] never touch a file; the frames go to outPath
 *   zref_bench cdict  <level> <dictPath> <recordsPath> <offsetsPath> <outPath> <threads> : the records compressed ONCE (one frame per record,
 *        CDict attached), the frames written back to back to outPath (bench.py hashes that stream: full-size parity of the records leg)
 *   zref_bench stream <totalBytes> <P%> <seed>      : RDG_genStdout to stdout (what `datagen -g -P -s` emits)
 *   zref_bench dict   <level> <dictPath> <recordsPath> <offsetsPath(u64 LE, nRec+1)> <seconds> <threads>
 *        one frame per record with ZSTD_createCDict + ZSTD_CCtx_refCDict + ZSTD_compress2 (the `zstd -b# -D dict` /
 *        contrib/largeNbDicts shape); threads split the records.
 */
#define ZSTD_STATIC_LINKING_ONLY
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <pthread.h>
#include "zstd.h"
#include "datagen.h"

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    int level; size_t chunk; const char* src; size_t n; char* dst; size_t dstCap; size_t csize; int err;
} job_t;

static void* worker(void* p)
{
    job_t* j = (job_t*)p;
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t off = 0, pos = 0;
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, j->level);
    if (getenv("ZREF_NOROW")) ZSTD_CCtx_setParameter(c, ZSTD_c_useRowMatchFinder, ZSTD_ps_disable);   /* hash-chain matcher */
    while (off < j->n) {
        size_t const len = j->n - off < j->chunk ? j->n - off : j->chunk;
        size_t r;
        if (getenv("ZREF_FRESH_CCTX")) {          /* a new CCtx per chunk: the row matcher's hash salt is then the same for every chunk
                                                     (a reused CCtx mixes the previous frames' hashes into it, zstd_compress.c:1964-1975) */
            ZSTD_freeCCtx(c); c = ZSTD_createCCtx();
            ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, j->level);
            if (getenv("ZREF_NOROW")) ZSTD_CCtx_setParameter(c, ZSTD_c_useRowMatchFinder, ZSTD_ps_disable);
        }
        r = ZSTD_compress2(c, j->dst + pos, j->dstCap - pos, j->src + off, len);
        if (ZSTD_isError(r)) { j->err = 1; break; }
        pos += r; off += len;
    }
    j->csize = pos;
    ZSTD_freeCCtx(c);
    return NULL;
}

static const char* g_file = NULL;

/* decode timing: frames[] laid out at k * bound in dst */
typedef struct { const char* comp; const size_t* csz; size_t bound; size_t k0, k1; char* out; size_t chunk; size_t n; int err; } ujob_t;
static void* uworker(void* p)
{
    ujob_t* j = (ujob_t*)p;
    ZSTD_DCtx* d = ZSTD_createDCtx();
    size_t k;
    for (k = j->k0; k < j->k1; k++) {
        size_t const want = (k + 1) * j->chunk <= j->n ? j->chunk : j->n - k * j->chunk;
        size_t const r = ZSTD_decompressDCtx(d, j->out + k * j->chunk, want, j->comp + k * j->bound, j->csz[k]);
        if (ZSTD_isError(r) || r != want) { j->err = 1; break; }
    }
    ZSTD_freeDCtx(d);
    return NULL;
}
static int dfile_main(char** argv)
{
    int const level = atoi(argv[2]); size_t const chunk = strtoull(argv[3], 0, 10); double const seconds = atof(argv[5]);
    int const T = atoi(argv[6]) > 0 ? atoi(argv[6]) : 1;
    FILE* f = fopen(argv[4], "rb"); long sz; char* src; char* comp; char* out; size_t* csz; size_t nChunks, bound, k, ctot = 0;
    ujob_t* jobs; pthread_t* th; double best = 1e30, t0; int runs = 0, t;
    ZSTD_CCtx* c = ZSTD_createCCtx();
    if (!f) { perror(argv[4]); return 1; }
    fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0, SEEK_SET);
    src = (char*)malloc((size_t)sz); out = (char*)malloc((size_t)sz + 64);
    if (!src || !out || fread(src, 1, (size_t)sz, f) != (size_t)sz) return 1;
    fclose(f);
    nChunks = ((size_t)sz + chunk - 1) / chunk; bound = ZSTD_compressBound(chunk);
    comp = (char*)malloc(bound * nChunks); csz = (size_t*)calloc(nChunks, sizeof(size_t));
    jobs = (ujob_t*)calloc((size_t)T, sizeof(ujob_t)); th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    if (!comp || !csz || !jobs || !th || !c) return 1;
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
    for (k = 0; k < nChunks; k++) {
        size_t const len = (k + 1) * chunk <= (size_t)sz ? chunk : (size_t)sz - k * chunk;
        csz[k] = ZSTD_compress2(c, comp + k * bound, bound, src + k * chunk, len);
        if (ZSTD_isError(csz[k])) return 1;
        ctot += csz[k];
    }
    ZSTD_freeCCtx(c);
    t0 = now_s();
    do {
        double const a = now_s();
        for (t = 0; t < T; t++) {
            jobs[t].comp = comp; jobs[t].csz = csz; jobs[t].bound = bound; jobs[t].k0 = nChunks * (size_t)t / (size_t)T; jobs[t].k1 = nChunks * (size_t)(t + 1) / (size_t)T;
            jobs[t].out = out; jobs[t].chunk = chunk; jobs[t].n = (size_t)sz; jobs[t].err = 0;
            if (T == 1) uworker(&jobs[t]); else pthread_create(&th[t], NULL, uworker, &jobs[t]);
        }
        for (t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); if (jobs[t].err) return 1; }
        {   double const d = now_s() - a; if (d < best) best = d; }
        runs++;
    } while (now_s() - t0 < seconds);
    if (memcmp(src, out, (size_t)sz)) return 1;
    printf("{\"level\": %d, \"chunk\": %zu, \"bytes\": %ld, \"csize\": %zu, \"ratio\": %.4f, \"best_s\": %.6f, \"MBps\": %.2f, \"runs\": %d, \"threads\": %d, \"mode\": \"decode\"}\n",
           level, chunk, sz, ctot, (double)sz / (double)ctot, best, (double)sz / best / 1e6, runs, T);
    return 0;
}

typedef struct { const ZSTD_CDict* cd; const char* src; const unsigned long long* offs; size_t r0, r1; char* dst; size_t dstCap; size_t csize; int err; } djob_t;
static void* dworker(void* p)
{
    djob_t* j = (djob_t*)p;
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t pos = 0, k;
    ZSTD_CCtx_refCDict(c, j->cd);
    for (k = j->r0; k < j->r1; k++) {
        size_t const r = ZSTD_compress2(c, j->dst + pos, j->dstCap - pos, j->src + j->offs[k], (size_t)(j->offs[k + 1] - j->offs[k]));
        if (ZSTD_isError(r)) { j->err = 1; break; }
        pos += r;
    }
    j->csize = pos;
    ZSTD_freeCCtx(c);
    return NULL;
}
static void* slurp(const char* path, size_t* n)
{
    FILE* f = fopen(path, "rb"); long sz; void* b;
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0, SEEK_SET);
    b = malloc((size_t)sz + 16);
    if (!b || fread(b, 1, (size_t)sz, f) != (size_t)sz) exit(1);
    fclose(f); *n = (size_t)sz;
    return b;
}
static int dict_main(char** argv)
{
    int const level = atoi(argv[2]);
    size_t dn, rn, on; double const seconds = atof(argv[6]); int const T = atoi(argv[7]) > 0 ? atoi(argv[7]) : 1;
    void* dict = slurp(argv[3], &dn); char* src = (char*)slurp(argv[4], &rn);
    unsigned long long* offs = (unsigned long long*)slurp(argv[5], &on);
    size_t const nRec = on / 8 - 1;
    ZSTD_CDict* cd = ZSTD_createCDict(dict, dn, level);
    size_t const cap = rn + rn / 128 + 128 * nRec + 1024;
    char* dst = (char*)malloc(cap);
    djob_t* jobs = (djob_t*)calloc((size_t)T, sizeof(djob_t));
    pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    double best = 1e30, t0 = now_s(); size_t csize = 0; int runs = 0, t;
    if (!cd || !dst) return 1;
    do {
        double const a = now_s(); size_t c0 = 0;
        for (t = 0; t < T; t++) {
            size_t const r0 = nRec * (size_t)t / (size_t)T, r1 = nRec * (size_t)(t + 1) / (size_t)T;
            size_t const b0 = (size_t)offs[r0] + (size_t)offs[r0] / 128 + 128 * r0;
            jobs[t].cd = cd; jobs[t].src = src; jobs[t].offs = offs; jobs[t].r0 = r0; jobs[t].r1 = r1;
            jobs[t].dst = dst + b0; jobs[t].dstCap = cap - b0;
            if (T == 1) dworker(&jobs[t]); else pthread_create(&th[t], NULL, dworker, &jobs[t]);
        }
        for (t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); if (jobs[t].err) return 1; c0 += jobs[t].csize; }
        {   double const d = now_s() - a; if (d < best) best = d; }
        csize = c0; runs++;
    } while (now_s() - t0 < seconds);
    printf("{\"level\": %d, \"records\": %zu, \"bytes\": %zu, \"csize\": %zu, \"ratio\": %.4f, \"best_s\": %.6f, \"MBps\": %.2f, \"runs\": %d, \"threads\": %d}\n",
           level, nRec, rn, csize, (double)rn / (double)csize, best, (double)rn / best / 1e6, runs, T);
    return 0;
}

/* ddict: the records compressed once (one frame each, CDict attached), then ZSTD_decompress_usingDDict per record is timed */
typedef struct { const ZSTD_DDict* dd; const char* comp; const size_t* coff; const unsigned long long* offs; size_t r0, r1; char* out; int err; } xjob_t;
static void* xworker(void* p)
{
    xjob_t* j = (xjob_t*)p;
    ZSTD_DCtx* d = ZSTD_createDCtx();
    size_t k;
    for (k = j->r0; k < j->r1; k++) {
        size_t const want = (size_t)(j->offs[k + 1] - j->offs[k]);
        size_t const r = ZSTD_decompress_usingDDict(d, j->out + j->offs[k], want, j->comp + j->coff[k], j->coff[k + 1] - j->coff[k], j->dd);
        if (ZSTD_isError(r) || r != want) { j->err = 1; break; }
    }
    ZSTD_freeDCtx(d);
    return NULL;
}
static int ddict_main(char** argv)
{
    int const level = atoi(argv[2]);
    size_t dn, rn, on; double const seconds = atof(argv[6]); int const T = atoi(argv[7]) > 0 ? atoi(argv[7]) : 1;
    void* dict = slurp(argv[3], &dn); char* src = (char*)slurp(argv[4], &rn);
    unsigned long long* offs = (unsigned long long*)slurp(argv[5], &on);
    size_t const nRec = on / 8 - 1;
    ZSTD_CDict* cd = ZSTD_createCDict(dict, dn, level);
    ZSTD_DDict* dd = ZSTD_createDDict(dict, dn);
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t const cap = rn + rn / 128 + 128 * nRec + 1024;
    char* comp = (char*)malloc(cap); char* out = (char*)malloc(rn + 64);
    size_t* coff = (size_t*)calloc(nRec + 1, sizeof(size_t));
    xjob_t* jobs = (xjob_t*)calloc((size_t)T, sizeof(xjob_t));
    pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    double best = 1e30, t0; int runs = 0, t; size_t k;
    if (!cd || !dd || !c || !comp || !out || !coff) return 1;
    ZSTD_CCtx_refCDict(c, cd);
    for (k = 0; k < nRec; k++) {
        size_t const r = ZSTD_compress2(c, comp + coff[k], cap - coff[k], src + offs[k], (size_t)(offs[k + 1] - offs[k]));
        if (ZSTD_isError(r)) return 1;
        coff[k + 1] = coff[k] + r;
    }
    t0 = now_s();
    do {
        double const a = now_s();
        for (t = 0; t < T; t++) {
            jobs[t].dd = dd; jobs[t].comp = comp; jobs[t].coff = coff; jobs[t].offs = offs; jobs[t].out = out; jobs[t].err = 0;
            jobs[t].r0 = nRec * (size_t)t / (size_t)T; jobs[t].r1 = nRec * (size_t)(t + 1) / (size_t)T;
            if (T == 1) xworker(&jobs[t]); else pthread_create(&th[t], NULL, xworker, &jobs[t]);
        }
        for (t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); if (jobs[t].err) return 1; }
        {   double const d = now_s() - a; if (d < best) best = d; }
        runs++;
    } while (now_s() - t0 < seconds);
    if (memcmp(src, out, rn)) return 1;
    printf("{\"level\": %d, \"records\": %zu, \"bytes\": %zu, \"csize\": %zu, \"best_s\": %.6f, \"MBps\": %.2f, \"runs\": %d, \"threads\": %d, \"mode\": \"decode\"}\n",
           level, nRec, rn, coff[nRec], best, (double)rn / best / 1e6, runs, T);
    return 0;
}

/* cfile: compress once with T threads, write the frames in chunk order */
static int cfile_main(char** argv)
{
    int const level = atoi(argv[2]); size_t const chunk = strtoull(argv[3], 0, 10);
    int const T = atoi(argv[6]) > 0 ? atoi(argv[6]) : 1;
    size_t total; char* src = (char*)slurp(argv[4], &total);
    size_t const nChunks = (total + chunk - 1) / chunk, bound = ZSTD_compressBound(chunk);
    char* dst = (char*)malloc(bound * (nChunks ? nChunks : 1));
    job_t* jobs = (job_t*)calloc((size_t)T, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    FILE* o; size_t csize = 0; int t; double const a = now_s();
    if (!dst || !jobs || !th) return 1;
    for (t = 0; t < T; t++) {
        size_t const k0 = nChunks * (size_t)t / (size_t)T, k1 = nChunks * (size_t)(t + 1) / (size_t)T;
        size_t const b0 = k0 * chunk, b1 = (k1 * chunk < total) ? k1 * chunk : total;
        jobs[t].level = level; jobs[t].chunk = chunk; jobs[t].src = src + b0; jobs[t].n = b1 - b0;
        jobs[t].dst = dst + k0 * bound; jobs[t].dstCap = (k1 - k0) * bound;
        if (T == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); if (jobs[t].err) return 1; }
    o = fopen(argv[5], "wb");
    if (!o) { perror(argv[5]); return 1; }
    for (t = 0; t < T; t++) { if (fwrite(jobs[t].dst, 1, jobs[t].csize, o) != jobs[t].csize) return 1; csize += jobs[t].csize; }
    fclose(o);
    printf("{\"level\": %d, \"chunk\": %zu, \"bytes\": %zu, \"csize\": %zu, \"seconds\": %.3f, \"threads\": %d}\n", level, chunk, total, csize, now_s() - a, T);
    return 0;
}

/* mtfile: one frame through the reference's job pool */
static int mtfile_main(char** argv)
{
    int const level = atoi(argv[2]), workers = atoi(argv[3]); int const jobSize = atoi(argv[6]);
    size_t total; char* src = (char*)slurp(argv[4], &total);
    size_t const bound = ZSTD_compressBound(total);
    char* dst = (char*)malloc(bound ? bound : 1);
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r; FILE* o; double a, b;
    if (!dst || !c) return 1;
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
    if (ZSTD_isError(ZSTD_CCtx_setParameter(c, ZSTD_c_nbWorkers, workers))) { fprintf(stderr, "reference built without ZSTD_MULTITHREAD\n"); return 1; }
    if (jobSize) ZSTD_CCtx_setParameter(c, ZSTD_c_jobSize, jobSize);
    a = now_s(); r = ZSTD_compress2(c, dst, bound, src, total); b = now_s();
    if (ZSTD_isError(r)) { fprintf(stderr, "%s\n", ZSTD_getErrorName(r)); return 1; }
    o = fopen(argv[5], "wb");
    if (!o) { perror(argv[5]); return 1; }
    if (fwrite(dst, 1, r, o) != r) return 1;
    fclose(o);
    printf("{\"level\": %d, \"workers\": %d, \"jobSize\": %d, \"bytes\": %zu, \"csize\": %zu, \"seconds\": %.4f, \"MBps\": %.2f}\n",
           level, workers, jobSize, total, r, b - a, (double)total / (b - a) / 1e6);
    ZSTD_freeCCtx(c);
    return 0;
}

/* cdict: the records compressed once with T threads, frames written in record order */
static int cdict_main(char** argv)
{
    int const level = atoi(argv[2]);
    size_t dn, rn, on; int const T = atoi(argv[7]) > 0 ? atoi(argv[7]) : 1;
    void* dict = slurp(argv[3], &dn); char* src = (char*)slurp(argv[4], &rn);
    unsigned long long* offs = (unsigned long long*)slurp(argv[5], &on);
    size_t const nRec = on / 8 - 1;
    ZSTD_CDict* cd = ZSTD_createCDict(dict, dn, level);
    size_t const cap = rn + rn / 128 + 128 * nRec + 1024;
    char* dst = (char*)malloc(cap);
    djob_t* jobs = (djob_t*)calloc((size_t)T, sizeof(djob_t));
    pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    FILE* o; size_t csize = 0; int t; double const a = now_s();
    if (!cd || !dst || !jobs || !th) return 1;
    for (t = 0; t < T; t++) {
        size_t const r0 = nRec * (size_t)t / (size_t)T, r1 = nRec * (size_t)(t + 1) / (size_t)T;
        size_t const b0 = (size_t)offs[r0] + (size_t)offs[r0] / 128 + 128 * r0;
        jobs[t].cd = cd; jobs[t].src = src; jobs[t].offs = offs; jobs[t].r0 = r0; jobs[t].r1 = r1;
        jobs[t].dst = dst + b0; jobs[t].dstCap = cap - b0;
        if (T == 1) dworker(&jobs[t]); else pthread_create(&th[t], NULL, dworker, &jobs[t]);
    }
    for (t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); if (jobs[t].err) return 1; }
    o = fopen(argv[6], "wb");
    if (!o) { perror(argv[6]); return 1; }
    for (t = 0; t < T; t++) { if (fwrite(jobs[t].dst, 1, jobs[t].csize, o) != jobs[t].csize) return 1; csize += jobs[t].csize; }
    fclose(o);
    printf("{\"level\": %d, \"records\": %zu, \"bytes\": %zu, \"csize\": %zu, \"seconds\": %.3f, \"threads\": %d}\n", level, nRec, rn, csize, now_s() - a, T);
    return 0;
}

/* ctile: cfile on a corpus tiled in memory (copy c starts at offset c*shift mod len and wraps around) */
static int ctile_main(char** argv)
{
    int const level = atoi(argv[2]); size_t const chunk = strtoull(argv[3], 0, 10);
    size_t const copies = strtoull(argv[5], 0, 10), shift = strtoull(argv[6], 0, 10), total = strtoull(argv[7], 0, 10);
    int const T = atoi(argv[9]) > 0 ? atoi(argv[9]) : 1;
    size_t L; char* base = (char*)slurp(argv[4], &L);
    char* src = (char*)malloc(total + 16);
    size_t const nChunks = (total + chunk - 1) / chunk, bound = ZSTD_compressBound(chunk);
    char* dst = (char*)malloc(bound * (nChunks ? nChunks : 1));
    job_t* jobs = (job_t*)calloc((size_t)T, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
    FILE* o; size_t csize = 0, pos = 0, c = 0; int t; double a;
    if (!src || !dst || !jobs || !th || !L) return 1;
    (void)copies;
    while (pos < total) {
        size_t const s0 = (c * shift) % L;
        size_t take = L - s0 < total - pos ? L - s0 : total - pos;
        memcpy(src + pos, base + s0, take); pos += take;
        take = s0 < total - pos ? s0 : total - pos;
        memcpy(src + pos, base, take); pos += take;
        c++;
    }
    a = now_s();
    for (t = 0; t < T; t++) {
        size_t const k0 = nChunks * (size_t)t / (size_t)T, k1 = nChunks * (size_t)(t + 1) / (size_t)T;
        size_t const b0 = k0 * chunk, b1 = (k1 * chunk < total) ? k1 * chunk : total;
        jobs[t].level = level; jobs[t].chunk = chunk; jobs[t].src = src + b0; jobs[t].n = b1 - b0;
        jobs[t].dst = dst + k0 * bound; jobs[t].dstCap = (k1 - k0) * bound;
        if (T == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    for (t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); if (jobs[t].err) return 1; }
    {   double const secs = now_s() - a;
        o = fopen(argv[8], "wb");
        if (!o) { perror(argv[8]); return 1; }
        for (t = 0; t < T; t++) { if (fwrite(jobs[t].dst, 1, jobs[t].csize, o) != jobs[t].csize) return 1; csize += jobs[t].csize; }
        fclose(o);
        printf("{\"level\": %d, \"chunk\": %zu, \"bytes\": %zu, \"csize\": %zu, \"seconds\": %.3f, \"MBps\": %.2f, \"threads\": %d}\n",
               level, chunk, total, csize, secs, (double)total / secs / 1e6, T);
    }
    return 0;
}

int main(int argc, char** argv)
{
    if (argc >= 10 && !strcmp(argv[1], "ctile")) return ctile_main(argv);
    if (argc >= 8 && !strcmp(argv[1], "cdict")) return cdict_main(argv);
    if (argc >= 7 && !strcmp(argv[1], "mtfile")) return mtfile_main(argv);
    if (argc >= 7 && !strcmp(argv[1], "cfile")) return cfile_main(argv);
    if (argc >= 8 && !strcmp(argv[1], "dict")) return dict_main(argv);
    if (argc >= 8 && !strcmp(argv[1], "ddict")) return ddict_main(argv);
    if (argc >= 7 && !strcmp(argv[1], "dfile")) return dfile_main(argv);
    if (argc >= 5 && !strcmp(argv[1], "stream")) {
        RDG_genStdout(strtoull(argv[2], 0, 10), atof(argv[3]) / 100.0, 0.0, (unsigned)atoi(argv[4]));
        return 0;
    }
    if (argc >= 7 && !strcmp(argv[1], "file")) {      /* rewrite argv into the bench form, input from file */
        static char* nv[9]; static char szbuf[32];
        FILE* f = fopen(argv[4], "rb"); long sz;
        if (!f) { perror(argv[4]); return 1; }
        fseek(f, 0, SEEK_END); sz = ftell(f); fclose(f);
        snprintf(szbuf, sizeof(szbuf), "%ld", sz);
        g_file = argv[4];
        nv[0] = argv[0]; nv[1] = (char*)"bench"; nv[2] = argv[2]; nv[3] = argv[3]; nv[4] = szbuf; nv[5] = (char*)"0"; nv[6] = (char*)"0";
        nv[7] = argv[5]; nv[8] = argv[6];
        argv = nv; argc = 9;
    }
    if (argc < 9 || strcmp(argv[1], "bench")) {
        fprintf(stderr, "usage: %s bench level chunk total P seed seconds threads | stream total P seed\n", argv[0]);
        return 2;
    }
    {   int const level = atoi(argv[2]);
        size_t const chunk = strtoull(argv[3], 0, 10);
        size_t const total = strtoull(argv[4], 0, 10);
        double const P = atof(argv[5]) / 100.0;
        unsigned const seed = (unsigned)atoi(argv[6]);
        double const seconds = atof(argv[7]);
        int const T = atoi(argv[8]) > 0 ? atoi(argv[8]) : 1;
        char* src = (char*)malloc(total);
        size_t const nChunks = (total + chunk - 1) / chunk;
        size_t const cap = ZSTD_compressBound(chunk) * nChunks;
        char* dst = (char*)malloc(cap);
        job_t* jobs = (job_t*)calloc((size_t)T, sizeof(job_t));
        pthread_t* th = (pthread_t*)calloc((size_t)T, sizeof(pthread_t));
        double best = 1e30, t0 = now_s(); size_t csize = 0; int runs = 0, t;
        if (!src || !dst || !jobs || !th) return 1;
        if (g_file) { FILE* f = fopen(g_file, "rb"); if (!f || fread(src, 1, total, f) != total) return 1; fclose(f); }
        else RDG_genBuffer(src, total, P, 0.0, seed);
        do {
            double const a = now_s();
            size_t c0 = 0;
            for (t = 0; t < T; t++) {
                size_t const k0 = nChunks * (size_t)t / (size_t)T, k1 = nChunks * (size_t)(t + 1) / (size_t)T;
                size_t const b0 = k0 * chunk, b1 = (k1 * chunk < total) ? k1 * chunk : total;
                jobs[t].level = level; jobs[t].chunk = chunk; jobs[t].src = src + b0; jobs[t].n = b1 - b0;
                jobs[t].dst = dst + k0 * ZSTD_compressBound(chunk); jobs[t].dstCap = (k1 - k0) * ZSTD_compressBound(chunk);
                if (T == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
            }
            for (t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); if (jobs[t].err) return 1; c0 += jobs[t].csize; }
            {   double const d = now_s() - a; if (d < best) best = d; }
            csize = c0; runs++;
        } while (now_s() - t0 < seconds);
        printf("{\"level\": %d, \"chunk\": %zu, \"bytes\": %zu, \"csize\": %zu, \"ratio\": %.4f, "
               "\"best_s\": %.6f, \"MBps\": %.2f, \"runs\": %d, \"threads\": %d}\n",
               level, chunk, total, csize, (double)total / (double)csize, best, (double)total / best / 1e6, runs, T);
        return 0;
    }
}
