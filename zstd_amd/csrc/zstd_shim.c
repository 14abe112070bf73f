/* zstd_shim.c — libzstd_hipshim.so: the reference's ZSTD_* names (include/zstd_hip_dropin.h) over the C ABI of
 * libzstd_hip.so (include/zstd_hip.h).  Plain C host code; the compression itself runs in the gfx950 kernels.
 * A ZSTD_CCtx owns one zhip_ctx (device = $ZHIP_DEVICE, default 0), created on first use and grown when a call
 * needs more 128 KB units than it holds. */
#include <stdlib.h>
#include <string.h>
#include "../../include/zstd_hip.h"
#include "../../include/zstd_hip_dropin.h"

#define SHIM_UNIT 131072u
#define SHIM_ERR(code) ((size_t)-(long)(code))
enum { E_GENERIC = 1, E_parameter_unsupported = 40, E_parameter_outOfBound = 42, E_stage_wrong = 60, E_memory_allocation = 64 };   /* lib/zstd_errors.h:60-101 */

struct ZSTD_CDict_s { zhip_cdict* d; };
struct ZSTD_CCtx_s {
    zhip_multi* zm;                  /* $ZHIP_DEVICES=0,1,...: sources are sharded over those devices (zhip_compress_multi) */
    zhip_ctx* z;
    size_t    zUnits;
    int       level;                 /* ZSTD_c_compressionLevel; 0 means default (3), lib/zstd.h:337-349 */
    int       checksum;              /* ZSTD_c_checksumFlag */
    int       rowMode;               /* ZSTD_c_useRowMatchFinder: 0 auto, 1 enable, 2 disable (lib/zstd.h ZSTD_paramSwitch_e) */
    unsigned  cp[7];                 /* ZSTD_c_windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy; 0 = the level's */
    int       singleFrame;           /* ZHIP_c_singleFrame / $ZHIP_SINGLE_FRAME: sources above 128 KB as ONE multi-block frame (zhip_compress_frames) */
    int       workers, jobSize, overlapLog;   /* ZSTD_c_nbWorkers / ZSTD_c_jobSize / ZSTD_c_overlapLog: sources above 512 KB as the frame the reference's job pool emits (zhip_compress_frames_mt) */
    const ZSTD_CDict* cdict;         /* ZSTD_CCtx_refCDict: sticky until reset / NULL (lib/zstd.h:1088-1102) */
};

static int shim_device(void) { const char* e = getenv("ZHIP_DEVICE"); return e ? atoi(e) : 0; }

ZSTD_CCtx* ZSTD_createCCtx(void)
{
    ZSTD_CCtx* c = (ZSTD_CCtx*)calloc(1, sizeof(*c));
    if (c) { const char* e = getenv("ZHIP_SINGLE_FRAME"); c->level = 3; c->singleFrame = e && atoi(e) != 0; }      /* ZSTD_CLEVEL_DEFAULT, lib/zstd.h:129 */
    return c;
}
size_t ZSTD_freeCCtx(ZSTD_CCtx* c)
{
    if (c) { if (c->z) zhip_destroy(c->z); if (c->zm) zhip_multi_destroy(c->zm); free(c); }
    return 0;
}
size_t ZSTD_CCtx_reset(ZSTD_CCtx* c, ZSTD_ResetDirective reset)
{
    if (!c) return SHIM_ERR(E_GENERIC);
    if (reset == ZSTD_reset_parameters || reset == ZSTD_reset_session_and_parameters) { c->level = 3; c->cdict = NULL; c->checksum = 0; c->rowMode = 0; c->workers = c->jobSize = c->overlapLog = 0; memset(c->cp, 0, sizeof(c->cp)); }
    return 0;
}
static size_t shim_set_cp(ZSTD_CCtx* c, int idx, int value, int lo, int hi)
{
    if (value != 0 && (value < lo || value > hi)) return SHIM_ERR(E_parameter_outOfBound);
    c->cp[idx] = (unsigned)value;
    return 0;
}
size_t ZSTD_CCtx_setParameter(ZSTD_CCtx* c, int param, int value)
{
    if (!c) return SHIM_ERR(E_GENERIC);
    switch (param) {
    case ZSTD_c_compressionLevel: c->level = value; return 0;     /* validated when used (clamping is the reference's behaviour for out-of-range levels; unsupported strategies fail at compress time) */
    /* advanced parameters (lib/compress/zstd_compress.c:710-768): kept on the CCtx, 0 = use the level's; bounds as
       ZSTD_cParam_getBounds (:433-470); what the device cannot run fails at compress time with parameter_unsupported */
    case ZSTD_c_windowLog:    return shim_set_cp(c, 0, value, 10, 31);
    case ZSTD_c_chainLog:     return shim_set_cp(c, 1, value, 6, 30);
    case ZSTD_c_hashLog:      return shim_set_cp(c, 2, value, 6, 30);
    case ZSTD_c_searchLog:    return shim_set_cp(c, 3, value, 1, 30);
    case ZSTD_c_minMatch:     return shim_set_cp(c, 4, value, 3, 7);
    case ZSTD_c_targetLength: return shim_set_cp(c, 5, value, 0, 131072);
    case ZSTD_c_strategy:     return shim_set_cp(c, 6, value, 1, 9);
    case ZSTD_c_useRowMatchFinder: if (value < 0 || value > 2) return SHIM_ERR(E_parameter_outOfBound); c->rowMode = value; return 0;
    case ZSTD_c_contentSizeFlag: return value == 1 ? 0 : SHIM_ERR(E_parameter_unsupported);
    case ZSTD_c_checksumFlag:    c->checksum = value != 0; return 0;
    case ZSTD_c_dictIDFlag:      return 0;                        /* no dictionary can be attached: the flag has no effect */
    /* multi-threaded mode (lib/zstd.h:452-491): the worker count does not change the reference's bytes, only that there are jobs; the
       device runs a workgroup per job whatever the count.  Bounds: ZSTDMT_NBWORKERS_MAX 256, job size 0 or up to 1 GiB (clamped from
       below to 512 KB at use, zstdmt_compress.c:1250), overlap log 0..9 */
    case ZSTD_c_nbWorkers:       if (value < 0 || value > 256) return SHIM_ERR(E_parameter_outOfBound); c->workers = value; return 0;
    case ZSTD_c_jobSize:         if (value < 0 || value > (1 << 30)) return SHIM_ERR(E_parameter_outOfBound); c->jobSize = value; return 0;
    case ZSTD_c_overlapLog:      if (value < 0 || value > 9) return SHIM_ERR(E_parameter_outOfBound); c->overlapLog = value; return 0;
    case ZHIP_c_singleFrame:     c->singleFrame = value != 0; return 0;
    default: return SHIM_ERR(E_parameter_unsupported);
    }
}

static size_t shim_ensure(ZSTD_CCtx* c, size_t units)
{
    if (!c->z || c->zUnits < units) {
        size_t want = units < 64 ? 64 : units;
        if (c->z) { zhip_destroy(c->z); c->z = NULL; }
        c->z = zhip_create(shim_device(), want);
        if (!c->z) return SHIM_ERR(E_memory_allocation);
        c->zUnits = want;
    }
    zhip_set_frame_checksum(c->z, c->checksum);
    zhip_set_row_matcher(c->z, c->rowMode ? c->rowMode : -1);     /* always pushed: ZSTD_ps_auto (also after a parameter reset) = the device context's own default ($ZHIP_ROW_MATCHER, else the reference's) */
    return 0;
}

/* one source compressed with an attached dictionary = one record of the device's records path (include/zstd_hip.h);
 * sources above the reference's attach cut-off take its copy path (k_parse_ext); above 128 KB -> parameter_unsupported */
static size_t shim_compress_cdict(ZSTD_CCtx* c, const ZSTD_CDict* cd, void* dst, size_t cap, const void* src, size_t n)
{
    unsigned long long offs[2];
    size_t need, r;
    if (!c || !cd || !cd->d) return SHIM_ERR(E_GENERIC);
    {   size_t const e = shim_ensure(c, 1); if (zhip_isError(e)) return e; }
    offs[0] = 0; offs[1] = n;
    need = zhip_records_bound(offs, 1);
    if (cap >= need) return zhip_compress_records(c->z, cd->d, dst, cap, src, offs, 1, NULL);
    {   void* tmp = malloc(need ? need : 1);
        if (!tmp) return SHIM_ERR(E_memory_allocation);
        r = zhip_compress_records(c->z, cd->d, tmp, need, src, offs, 1, NULL);
        if (!zhip_isError(r)) { if (r <= cap) memcpy(dst, tmp, r); else r = SHIM_ERR(70 /* dstSize_tooSmall */); }
        free(tmp);
        return r;
    }
}

/* $ZHIP_DEVICES: comma-separated device ordinals -> the multi-device host path (created on first use).  Without it, a source of SHIM_LANES_MIN bytes or more that is to
 * become the frame-per-128-KB stream takes the same path on the CCtx's ONE device: its lanes overlap the staging copies, the PCIe transfers and the kernels of
 * consecutive 128 MB chunks, where the plain call does them one after the other on one stream (1 GiB of datagen at level 1: 35 vs 8 GB/s, bench.py `end_to_end`).  The
 * bytes are the same either way (units are independent).  Smaller sources stay on the plain call: the lanes' pinned staging is 1 GB. */
#define SHIM_LANES_MIN ((size_t)256 << 20)
static zhip_multi* shim_multi(ZSTD_CCtx* c, size_t srcSize)
{
    const char* e = getenv("ZHIP_DEVICES");
    int dev[64]; int n = 0;
    if (c->zm) return c->zm;
    if (e && *e) { while (*e && n < 64) { dev[n++] = atoi(e); while (*e && *e != ',') e++; if (*e == ',') e++; } }
    else if (srcSize >= SHIM_LANES_MIN && !c->singleFrame && c->workers == 0) dev[n++] = shim_device();
    if (n) c->zm = zhip_multi_create(dev, n, 0);
    return c->zm;
}

static size_t shim_compress(ZSTD_CCtx* c, void* dst, size_t cap, const void* src, size_t n, int level)
{
    size_t const units = n ? (n + SHIM_UNIT - 1) / SHIM_UNIT : 1;
    size_t frameNeed = 0;
    if (!c) return SHIM_ERR(E_GENERIC);
    if (level == 0) level = 3;
    if (units > 1 && (getenv("ZHIP_DEVICES") ? 1 : (n >= SHIM_LANES_MIN && !c->singleFrame && c->workers == 0)) && shim_multi(c, n)) {
        /* big sources: sharded over $ZHIP_DEVICES — or, from SHIM_LANES_MIN bytes on, over the lanes of the CCtx's one device — with overlapped copies; gathers straight into dst */
        zhip_multi_set_frame_checksum(c->zm, c->checksum);
        zhip_multi_set_row_matcher(c->zm, c->rowMode ? c->rowMode : -1);
        if (c->workers > 0 && n > (512u << 10)) {    /* ZSTD_c_nbWorkers: one frame, its jobs spread over the lanes */
            size_t const r = zhip_compress_frame_mt_multi(c->zm, dst, cap, src, n, level, c->cp, (size_t)c->jobSize, c->overlapLog);
            if (!zhip_isError(r) || r != SHIM_ERR(E_parameter_unsupported)) return r;
        }
        return zhip_compress_multi(c->zm, dst, cap, src, n, level, c->cp, SHIM_UNIT, NULL);
    }
    {   size_t const e = shim_ensure(c, units); if (zhip_isError(e)) return e; }
    {   unsigned long long const fo[2] = { 0, n };
        frameNeed = zhip_frames_bound(fo, 1);           /* = ZSTD_COMPRESSBOUND(n): what a caller sizing dst with the reference's macro provides */
    }
    if (c->workers > 0 && n > (512u << 10) && cap >= frameNeed) {
        /* ZSTD_c_nbWorkers >= 1: the frame of the reference's job pool, a workgroup per job.  At or below 512 KB the reference drops
           the workers (zstd_compress.c:6215) and so does this; unsupported strategies fall through like below */
        unsigned long long const offs[2] = { 0, n };
        size_t const r = zhip_compress_frames_mt(c->z, dst, cap, src, offs, 1, level, c->cp, (size_t)c->jobSize, c->overlapLog, NULL);
        if (!zhip_isError(r) || r != SHIM_ERR(E_parameter_unsupported)) return r;
    }
    if (units > 1 && c->singleFrame && cap >= frameNeed) {
        /* the reference's own output shape: one frame, many blocks.  Strategies the frame kernel does not run
           (parameter_unsupported) fall through to the frame-per-128-KB stream below */
        unsigned long long const offs[2] = { 0, n };
        size_t const r = zhip_compress_frames(c->z, dst, cap, src, offs, 1, level, c->cp, NULL);
        if (!zhip_isError(r) || r != SHIM_ERR(E_parameter_unsupported)) return r;
    }
    /* zhip_compress wants room for its own bound; the reference only needs ZSTD_compressBound(n) for a guaranteed
       success and otherwise tries — give the device a private bounce buffer when the caller's is smaller */
    {   size_t const need = zhip_compressBound(n, SHIM_UNIT);
        if (cap >= need) return zhip_compress_params(c->z, dst, cap, src, n, level, c->cp, SHIM_UNIT, NULL);
        {   void* tmp = malloc(need ? need : 1);
            size_t r;
            if (!tmp) return SHIM_ERR(E_memory_allocation);
            r = zhip_compress_params(c->z, tmp, need, src, n, level, c->cp, SHIM_UNIT, NULL);
            if (!zhip_isError(r)) { if (r <= cap) memcpy(dst, tmp, r); else r = SHIM_ERR(70 /* dstSize_tooSmall */); }
            free(tmp);
            return r;
        }
    }
}

size_t ZSTD_compress2(ZSTD_CCtx* c, void* dst, size_t cap, const void* src, size_t n)
{
    if (c && c->cdict) return shim_compress_cdict(c, c->cdict, dst, cap, src, n);   /* the CDict's level takes priority (zstd_compress.c:6276-6282) */
    return shim_compress(c, dst, cap, src, n, c ? c->level : 3);
}

/* ---- the streaming entry point, for the one case the reference itself handles as a single ZSTD_compress2-like pass
 * (zstd_compress.c:6069-6084): the first call of a frame says ZSTD_e_end, all input is there and the output has room for
 * ZSTD_compressBound of it.  Then the bytes are ZSTD_compress2's.  Anything that would need the streaming state machine
 * (ZSTD_e_continue / ZSTD_e_flush with input, too little output room) returns parameter_unsupported: it stays with the reference. */
size_t ZSTD_compressStream2(ZSTD_CCtx* c, ZSTD_outBuffer* out, ZSTD_inBuffer* in, ZSTD_EndDirective endOp)
{
    size_t n, room, r;
    if (!c || !out || !in || out->pos > out->size || in->pos > in->size) return SHIM_ERR(E_GENERIC);
    n = in->size - in->pos; room = out->size - out->pos;
    if (endOp != ZSTD_e_end) return (n == 0 && endOp == ZSTD_e_flush) ? 0 : SHIM_ERR(E_parameter_unsupported);
    if (room < ZSTD_compressBound(n)) return SHIM_ERR(E_parameter_unsupported);
    r = ZSTD_compress2(c, (char*)out->dst + out->pos, room, (const char*)in->src + in->pos, n);
    if (zhip_isError(r)) return r;
    in->pos = in->size; out->pos += r;
    return 0;                                                            /* frame completely written */
}
ZSTD_CStream* ZSTD_createCStream(void) { return ZSTD_createCCtx(); }     /* lib/zstd.h:756-760: a CStream is a CCtx */
size_t ZSTD_freeCStream(ZSTD_CStream* zcs) { return ZSTD_freeCCtx(zcs); }
size_t ZSTD_initCStream(ZSTD_CStream* zcs, int level)                    /* :842 = reset session + set level */
{
    if (!zcs) return SHIM_ERR(E_GENERIC);
    zcs->cdict = NULL;
    return ZSTD_CCtx_setParameter(zcs, ZSTD_c_compressionLevel, level);
}
size_t ZSTD_CStreamInSize(void) { return SHIM_UNIT; }                    /* :822-823 */
size_t ZSTD_CStreamOutSize(void) { return ZSTD_compressBound(SHIM_UNIT); }

/* ---- dictionaries (lib/zstd.h:979-995, :1102) */
ZSTD_CDict* ZSTD_createCDict(const void* dict, size_t dictSize, int level)
{
    ZSTD_CDict* cd = (ZSTD_CDict*)calloc(1, sizeof(*cd));
    if (!cd) return NULL;
    cd->d = zhip_create_cdict(shim_device(), dict, dictSize, level);
    if (!cd->d) { free(cd); return NULL; }
    return cd;
}
size_t ZSTD_freeCDict(ZSTD_CDict* cd)
{
    if (cd) { zhip_free_cdict(cd->d); free(cd); }
    return 0;
}
size_t ZSTD_CCtx_refCDict(ZSTD_CCtx* c, const ZSTD_CDict* cd)
{
    if (!c) return SHIM_ERR(E_GENERIC);
    c->cdict = cd;
    return 0;
}
size_t ZSTD_compress_usingCDict(ZSTD_CCtx* c, void* dst, size_t cap, const void* src, size_t n, const ZSTD_CDict* cd)
{
    return shim_compress_cdict(c, cd, dst, cap, src, n);
}
size_t ZSTD_compressCCtx(ZSTD_CCtx* c, void* dst, size_t cap, const void* src, size_t n, int level)
{   /* ignores the cctx's parameters (checksum flag included), like the reference (zstd_compress.c:5428) */
    int const ck = c ? c->checksum : 0, wk = c ? c->workers : 0; size_t r; unsigned cp[7];
    if (c) { c->checksum = 0; c->workers = 0; memcpy(cp, c->cp, sizeof(cp)); memset(c->cp, 0, sizeof(c->cp)); }
    r = shim_compress(c, dst, cap, src, n, level);
    if (c) { c->checksum = ck; c->workers = wk; memcpy(c->cp, cp, sizeof(cp)); }
    return r;
}
size_t ZSTD_compress(void* dst, size_t cap, const void* src, size_t n, int level)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r;
    if (!c) return SHIM_ERR(E_memory_allocation);
    r = shim_compress(c, dst, cap, src, n, level);
    ZSTD_freeCCtx(c);
    return r;
}

/* ---- decompression (lib/zstd.h:168-294, :998-1013): host buffers through zhip_decompress */
struct ZSTD_DCtx_s { zhip_dctx* z; };
struct ZSTD_DDict_s { zhip_ddict* d; };
ZSTD_DCtx* ZSTD_createDCtx(void) { return (ZSTD_DCtx*)calloc(1, sizeof(ZSTD_DCtx)); }
size_t ZSTD_freeDCtx(ZSTD_DCtx* d) { if (d) { if (d->z) zhip_free_dctx(d->z); free(d); } return 0; }
static size_t shim_decompress(ZSTD_DCtx* d, void* dst, size_t cap, const void* src, size_t n, const ZSTD_DDict* dd)
{
    if (!d) return SHIM_ERR(E_GENERIC);
    if (!d->z) { d->z = zhip_create_dctx(shim_device()); if (!d->z) return SHIM_ERR(E_memory_allocation); }
    return zhip_decompress(d->z, dd ? dd->d : NULL, dst, cap, src, n);
}
size_t ZSTD_decompressDCtx(ZSTD_DCtx* d, void* dst, size_t cap, const void* src, size_t n) { return shim_decompress(d, dst, cap, src, n, NULL); }
size_t ZSTD_decompress_usingDDict(ZSTD_DCtx* d, void* dst, size_t cap, const void* src, size_t n, const ZSTD_DDict* dd) { return shim_decompress(d, dst, cap, src, n, dd); }
size_t ZSTD_decompress(void* dst, size_t cap, const void* src, size_t n)
{
    ZSTD_DCtx* d = ZSTD_createDCtx(); size_t r;
    if (!d) return SHIM_ERR(E_memory_allocation);
    r = shim_decompress(d, dst, cap, src, n, NULL);
    ZSTD_freeDCtx(d);
    return r;
}
ZSTD_DDict* ZSTD_createDDict(const void* dict, size_t dictSize)
{
    ZSTD_DDict* dd = (ZSTD_DDict*)calloc(1, sizeof(*dd));
    if (!dd) return NULL;
    dd->d = zhip_create_ddict(shim_device(), dict, dictSize);
    if (!dd->d) { free(dd); return NULL; }
    return dd;
}
size_t ZSTD_freeDDict(ZSTD_DDict* dd) { if (dd) { zhip_free_ddict(dd->d); free(dd); } return 0; }
unsigned ZSTD_getDictID_fromDDict(const ZSTD_DDict* dd) { return dd ? zhip_ddict_id(dd->d) : 0; }
unsigned long long ZSTD_getFrameContentSize(const void* src, size_t n)
{   /* first frame only, like the reference (zstd_decompress.c:630-650); needs the header, not the whole frame */
    unsigned long long content = 0; const unsigned char* p = (const unsigned char*)src;
    static const unsigned did[4] = { 0, 1, 2, 4 }, fcsB[4] = { 0, 2, 4, 8 };
    unsigned fhd, single, fcsCode, nb, i; size_t pos;
    if (n < 5 || !(p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD)) return ZSTD_CONTENTSIZE_ERROR;
    fhd = p[4]; single = (fhd >> 5) & 1; fcsCode = fhd >> 6;
    pos = 5 + !single + did[fhd & 3]; nb = fcsB[fcsCode] + (single && !fcsCode);
    if (pos + nb > n) return ZSTD_CONTENTSIZE_ERROR;
    if (!nb) return ZSTD_CONTENTSIZE_UNKNOWN;
    for (i = 0; i < nb; i++) content |= (unsigned long long)p[pos + i] << (8 * i);
    return fcsCode == 1 ? content + 256 : content;
}
size_t ZSTD_findFrameCompressedSize(const void* src, size_t n) { return zhip_frame_compressed_size(src, n); }
unsigned long long ZSTD_findDecompressedSize(const void* src, size_t n)
{
    size_t const k = zhip_find_frames(src, n, NULL, NULL, NULL, NULL, 0); unsigned long long total = 0, *cs; size_t i;
    if (zhip_isError(k)) return ZSTD_CONTENTSIZE_ERROR;
    if (!k) return 0;
    cs = (unsigned long long*)malloc(k * sizeof(*cs));
    if (!cs) return ZSTD_CONTENTSIZE_ERROR;
    (void)zhip_find_frames(src, n, NULL, NULL, cs, NULL, k);
    for (i = 0; i < k; i++) { if (cs[i] == ZSTD_CONTENTSIZE_UNKNOWN) { free(cs); return ZSTD_CONTENTSIZE_UNKNOWN; } total += cs[i]; }
    free(cs);
    return total;
}
unsigned long long ZSTD_decompressBound(const void* src, size_t n)
{   /* stated content sizes where present, nbBlocks * blockSizeMax otherwise (zstd_decompress.c:ZSTD_decompressBound) */
    size_t const k = zhip_find_frames(src, n, NULL, NULL, NULL, NULL, 0); unsigned long long total = 0, *cs, *cb; size_t i;
    if (zhip_isError(k)) return ZSTD_CONTENTSIZE_ERROR;
    if (!k) return 0;
    cs = (unsigned long long*)malloc(2 * k * sizeof(*cs));
    if (!cs) return ZSTD_CONTENTSIZE_ERROR;
    cb = cs + k;
    (void)zhip_find_frames(src, n, NULL, NULL, cs, cb, k);
    for (i = 0; i < k; i++) total += cs[i] != ZSTD_CONTENTSIZE_UNKNOWN ? cs[i] : cb[i];
    free(cs);
    return total;
}
unsigned ZSTD_isFrame(const void* buffer, size_t size)
{
    const unsigned char* p = (const unsigned char*)buffer;
    if (size < 4) return 0;
    if (p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD) return 1;
    return (p[0] & 0xF0) == 0x50 && p[1] == 0x2A && p[2] == 0x4D && p[3] == 0x18;                  /* 0x184D2A5? skippable */
}
unsigned ZSTD_getDictID_fromFrame(const void* src, size_t n)
{
    const unsigned char* p = (const unsigned char*)src; unsigned fhd, single, code, id = 0, i; size_t pos;
    static const unsigned did[4] = { 0, 1, 2, 4 };
    if (n < 5 || !(p[0] == 0x28 && p[1] == 0xB5 && p[2] == 0x2F && p[3] == 0xFD)) return 0;
    fhd = p[4]; single = (fhd >> 5) & 1; code = fhd & 3; pos = 5 + !single;
    if (pos + did[code] > n) return 0;
    for (i = 0; i < did[code]; i++) id |= (unsigned)p[pos + i] << (8 * i);
    return id;
}
size_t ZSTD_compressBound(size_t n) { return zhip_compressBound(n, SHIM_UNIT); }
unsigned ZSTD_isError(size_t code) { return zhip_isError(code); }
const char* ZSTD_getErrorName(size_t code) { return zhip_getErrorName(code); }
int ZSTD_minCLevel(void) { return -131072; }                      /* -ZSTD_TARGETLENGTH_MAX, lib/zstd.h:1249 */
int ZSTD_maxCLevel(void) { return 10; }                           /* fast, dfast and the hash-chain greedy/lazy/lazy2 rows of units <= 128 KB; levels >= 5 reproduce the
                                                                      reference with ZSTD_c_useRowMatchFinder = ZSTD_ps_disable; small units at 9-10 (btlazy2) -> parameter_unsupported */
int ZSTD_defaultCLevel(void) { return 3; }
