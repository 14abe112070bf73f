/* examples/roundtrip.c — a program written against zstd.h, linked with -lzstd_hipshim instead of -lzstd.
 *
 * Only the reference's own names are used (ZSTD_compress2, ZSTD_decompress, ZSTD_createCDict, ...); with
 * -DUSE_REFERENCE_HEADER the reference's lib/zstd.h is included unchanged, otherwise include/zstd_hip_dropin.h (the same
 * prototypes).  Compression and decompression both run on the MI355X; the output is checked against the input.
 *
 *   gcc -O2 -Iinclude examples/roundtrip.c -Lzstd_amd -lzstd_hipshim -lzstd_hip -Wl,-rpath,$PWD/zstd_amd -o roundtrip
 *   ./roundtrip [bytes] [level]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef USE_REFERENCE_HEADER
#define ZSTD_STATIC_LINKING_ONLY          /* ZSTD_findDecompressedSize lives in the experimental section of lib/zstd.h */
#include "zstd.h"
#else
#include "zstd_hip_dropin.h"
#endif

static void fill(unsigned char* p, size_t n)
{   /* compressible: a small vocabulary of words */
    static const char* words[] = { "block ", "frame ", "literal ", "sequence ", "offset ", "window ", "huffman ", "entropy ", "match ", "hash " };
    size_t pos = 0; unsigned s = 12345;
    while (pos < n) {
        const char* w = words[(s = s * 1103515245u + 12345u) >> 16 & 7];
        size_t const l = strlen(w), k = l < n - pos ? l : n - pos;
        memcpy(p + pos, w, k); pos += k;
    }
}

int main(int argc, char** argv)
{
    size_t const n = argc > 1 ? (size_t)strtoull(argv[1], 0, 10) : (size_t)5 << 20;
    int const level = argc > 2 ? atoi(argv[2]) : 3;
    unsigned char* src = (unsigned char*)malloc(n ? n : 1);
    size_t const bound = ZSTD_compressBound(n);
    unsigned char* comp = (unsigned char*)malloc(bound);
    unsigned char* back = (unsigned char*)malloc(n ? n : 1);
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t cs, ds;
    if (!src || !comp || !back || !c) return 2;
    fill(src, n);
    ZSTD_CCtx_setParameter(c, ZSTD_c_compressionLevel, level);
    ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, 1);
    cs = ZSTD_compress2(c, comp, bound, src, n);
    if (ZSTD_isError(cs)) { fprintf(stderr, "compress: %s\n", ZSTD_getErrorName(cs)); return 1; }
    if (ZSTD_findDecompressedSize(comp, cs) != n) { fprintf(stderr, "findDecompressedSize mismatch\n"); return 1; }
    ds = ZSTD_decompress(back, n, comp, cs);
    if (ZSTD_isError(ds) || ds != n || memcmp(src, back, n)) { fprintf(stderr, "decompress: %s\n", ZSTD_isError(ds) ? ZSTD_getErrorName(ds) : "content differs"); return 1; }
    comp[cs / 2] ^= 0x10;                                     /* corrupt one bit: the decoder must refuse (checksum or structure) */
    ds = ZSTD_decompress(back, n, comp, cs);
    if (n > 64 && !ZSTD_isError(ds)) { fprintf(stderr, "corrupted stream was accepted\n"); return 1; }
    if (n > ((size_t)1 << 20)) {
        /* ZSTD_c_nbWorkers: the source as ONE standard frame (the bytes of the reference's job pool; a workgroup per job on the device),
         * and the streaming entry point in its one-shot form, which must give the same bytes */
        unsigned char* comp2 = (unsigned char*)malloc(bound);
        ZSTD_inBuffer in; ZSTD_outBuffer out;
        size_t r;
        if (!comp2) return 2;
        ZSTD_CCtx_setParameter(c, ZSTD_c_checksumFlag, 0);
        if (ZSTD_isError(ZSTD_CCtx_setParameter(c, ZSTD_c_nbWorkers, 4))) { fprintf(stderr, "ZSTD_c_nbWorkers refused\n"); return 1; }
        ZSTD_CCtx_setParameter(c, ZSTD_c_jobSize, 1 << 20);
        cs = ZSTD_compress2(c, comp, bound, src, n);
        if (ZSTD_isError(cs)) { fprintf(stderr, "compress2 with workers: %s\n", ZSTD_getErrorName(cs)); return 1; }
        if (level >= 1 && level <= 3 && ZSTD_getFrameContentSize(comp, cs) != n) { fprintf(stderr, "not a single frame\n"); return 1; }
        ds = ZSTD_decompress(back, n, comp, cs);
        if (ZSTD_isError(ds) || ds != n || memcmp(src, back, n)) { fprintf(stderr, "round trip of the job-pool frame failed\n"); return 1; }
        in.src = src; in.size = n; in.pos = 0; out.dst = comp2; out.size = bound; out.pos = 0;
        r = ZSTD_compressStream2(c, &out, &in, ZSTD_e_end);
        if (r != 0 || in.pos != n || out.pos != cs || memcmp(comp, comp2, cs)) { fprintf(stderr, "compressStream2(e_end) differs from compress2\n"); return 1; }
        printf("job-pool frame: %zu -> %zu bytes\n", n, cs);
        ZSTD_CCtx_setParameter(c, ZSTD_c_nbWorkers, 0);
        free(comp2);
    }
    {   /* dictionary round trip: ZSTD_createCDict / ZSTD_compress_usingCDict / ZSTD_createDDict / ZSTD_decompress_usingDDict */
        size_t const dn = n < 60000 ? n : 60000, rn = n < 3000 ? n : 3000;
        ZSTD_CDict* cd = ZSTD_createCDict(src, dn, 3);
        ZSTD_DDict* dd = ZSTD_createDDict(src, dn);
        ZSTD_DCtx* d = ZSTD_createDCtx();
        if (cd && dd && d && rn > 8) {
            cs = ZSTD_compress_usingCDict(c, comp, bound, src + dn / 2, rn, cd);
            if (ZSTD_isError(cs)) { fprintf(stderr, "compress_usingCDict: %s\n", ZSTD_getErrorName(cs)); return 1; }
            ds = ZSTD_decompress_usingDDict(d, back, rn, comp, cs, dd);
            if (ZSTD_isError(ds) || ds != rn || memcmp(src + dn / 2, back, rn)) { fprintf(stderr, "dictionary round trip failed\n"); return 1; }
            printf("dictionary record: %zu -> %zu bytes\n", rn, cs);
        }
        ZSTD_freeCDict(cd); ZSTD_freeDDict(dd); ZSTD_freeDCtx(d);
    }
    printf("roundtrip ok: %zu bytes, level %d\n", n, level);
    ZSTD_freeCCtx(c); free(src); free(comp); free(back);
    return 0;
}
