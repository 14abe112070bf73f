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

struct ZSTD_CCtx_s {
    zhip_ctx* z;
    size_t    zUnits;
    int       level;                 /* ZSTD_c_compressionLevel; 0 means default (3), lib/zstd.h:337-349 */
};

static int shim_device(void) { const char* e = getenv("ZHIP_DEVICE"); return e ? atoi(e) : 0; }

ZSTD_CCtx* ZSTD_createCCtx(void)
{
    ZSTD_CCtx* c = (ZSTD_CCtx*)calloc(1, sizeof(*c));
    if (c) c->level = 3;                                          /* ZSTD_CLEVEL_DEFAULT, lib/zstd.h:129 */
    return c;
}
size_t ZSTD_freeCCtx(ZSTD_CCtx* c)
{
    if (c) { if (c->z) zhip_destroy(c->z); free(c); }
    return 0;
}
size_t ZSTD_CCtx_reset(ZSTD_CCtx* c, ZSTD_ResetDirective reset)
{
    if (!c) return SHIM_ERR(E_GENERIC);
    if (reset == ZSTD_reset_parameters || reset == ZSTD_reset_session_and_parameters) c->level = 3;
    return 0;
}
size_t ZSTD_CCtx_setParameter(ZSTD_CCtx* c, int param, int value)
{
    if (!c) return SHIM_ERR(E_GENERIC);
    switch (param) {
    case ZSTD_c_compressionLevel: c->level = value; return 0;     /* validated when used (clamping is the reference's behaviour for out-of-range levels; unsupported strategies fail at compress time) */
    /* advanced parameters: only "0 = use the level's default" is representable on the device */
    case ZSTD_c_windowLog: case ZSTD_c_hashLog: case ZSTD_c_chainLog: case ZSTD_c_searchLog:
    case ZSTD_c_minMatch: case ZSTD_c_targetLength: case ZSTD_c_strategy:
        return value == 0 ? 0 : SHIM_ERR(E_parameter_unsupported);
    case ZSTD_c_contentSizeFlag: return value == 1 ? 0 : SHIM_ERR(E_parameter_unsupported);
    case ZSTD_c_checksumFlag:    return value == 0 ? 0 : SHIM_ERR(E_parameter_unsupported);
    case ZSTD_c_dictIDFlag:      return 0;                        /* no dictionary can be attached: the flag has no effect */
    case ZSTD_c_nbWorkers:       return value == 0 ? 0 : SHIM_ERR(E_parameter_unsupported);
    default: return SHIM_ERR(E_parameter_unsupported);
    }
}

static size_t shim_compress(ZSTD_CCtx* c, void* dst, size_t cap, const void* src, size_t n, int level)
{
    size_t const units = n ? (n + SHIM_UNIT - 1) / SHIM_UNIT : 1;
    if (!c) return SHIM_ERR(E_GENERIC);
    if (level == 0) level = 3;
    if (!c->z || c->zUnits < units) {
        size_t want = units < 64 ? 64 : units;
        if (c->z) { zhip_destroy(c->z); c->z = NULL; }
        c->z = zhip_create(shim_device(), want);
        if (!c->z) return SHIM_ERR(E_memory_allocation);
        c->zUnits = want;
    }
    /* zhip_compress wants room for its own bound; the reference only needs ZSTD_compressBound(n) for a guaranteed
       success and otherwise tries — give the device a private bounce buffer when the caller's is smaller */
    {   size_t const need = zhip_compressBound(n, SHIM_UNIT);
        if (cap >= need) return zhip_compress(c->z, dst, cap, src, n, level, SHIM_UNIT, NULL);
        {   void* tmp = malloc(need ? need : 1);
            size_t r;
            if (!tmp) return SHIM_ERR(E_memory_allocation);
            r = zhip_compress(c->z, tmp, need, src, n, level, SHIM_UNIT, NULL);
            if (!zhip_isError(r)) { if (r <= cap) memcpy(dst, tmp, r); else r = SHIM_ERR(70 /* dstSize_tooSmall */); }
            free(tmp);
            return r;
        }
    }
}

size_t ZSTD_compress2(ZSTD_CCtx* c, void* dst, size_t cap, const void* src, size_t n) { return shim_compress(c, dst, cap, src, n, c ? c->level : 3); }
size_t ZSTD_compressCCtx(ZSTD_CCtx* c, void* dst, size_t cap, const void* src, size_t n, int level) { return shim_compress(c, dst, cap, src, n, level); }   /* ignores the cctx's parameters, like the reference (zstd_compress.c:5428) */
size_t ZSTD_compress(void* dst, size_t cap, const void* src, size_t n, int level)
{
    ZSTD_CCtx* c = ZSTD_createCCtx();
    size_t r;
    if (!c) return SHIM_ERR(E_memory_allocation);
    r = shim_compress(c, dst, cap, src, n, level);
    ZSTD_freeCCtx(c);
    return r;
}
size_t ZSTD_compressBound(size_t n) { return zhip_compressBound(n, SHIM_UNIT); }
unsigned ZSTD_isError(size_t code) { return zhip_isError(code); }
const char* ZSTD_getErrorName(size_t code) { return zhip_getErrorName(code); }
int ZSTD_minCLevel(void) { return -131072; }                      /* -ZSTD_TARGETLENGTH_MAX, lib/zstd.h:1269 */
int ZSTD_maxCLevel(void) { return 4; }                            /* fast + dfast rows; level 4 below 16 KB is greedy and fails with parameter_unsupported */
int ZSTD_defaultCLevel(void) { return 3; }
