// zhip_host.h — host-only helpers of libzstd_hip: parameter selection and bounds.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace zhip {

struct CParams { unsigned windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy; };

// Level -> parameters for the strategies this library implements (fast = 1, dfast = 2).
// Values are the reference's rows (lib/compress/clevels.h:24-130: "base for negative levels", levels 1..4) for
// its four source-size classes; the selection + adjustment logic mirrors ZSTD_getCParams_internal
// (lib/compress/zstd_compress.c:7123-7145) and ZSTD_adjustCParams_internal (:1466-1602) for a known source size
// and no dictionary.  tests/test_host_params.py sweeps this against the real reference.
static inline bool host_get_cparams(int level, unsigned long long srcSize, CParams* out)
{
    static const CParams rows[4][5] = {
        /* srcSize > 256 KB */ {{19,12,13,1,6,1,1},{19,13,14,1,7,0,1},{20,15,16,1,6,0,1},{21,16,17,1,5,0,2},{21,18,18,1,5,0,2}},
        /* <= 256 KB       */ {{18,12,13,1,5,1,1},{18,13,14,1,6,0,1},{18,14,14,1,5,0,2},{18,16,16,1,4,0,2},{18,16,17,3,5,2,3}},
        /* <= 128 KB       */ {{17,12,12,1,5,1,1},{17,12,13,1,6,0,1},{17,13,15,1,5,0,1},{17,15,16,2,5,0,2},{17,17,17,2,4,0,2}},
        /* <= 16 KB        */ {{14,12,13,1,5,1,1},{14,14,15,1,5,0,1},{14,14,15,1,4,0,1},{14,14,15,2,4,0,2},{14,14,14,4,4,2,3}},
    };
    unsigned const cls = (srcSize <= 256u * 1024) + (srcSize <= 128u * 1024) + (srcSize <= 16u * 1024);
    int row = level == 0 ? 3 : (level < 0 ? 0 : level);
    if (row > 4) return false;
    CParams cp = rows[cls][row];
    if (cp.strategy > 2) return false;
    if (level < 0) { long const lv = level < -131072 ? -131072 : level; cp.targetLength = (unsigned)(-lv); }
    if (srcSize <= (1ULL << 30)) {
        uint32_t const t = (uint32_t)srcSize;
        unsigned const srcLog = t < 64 ? 6 : 32 - (unsigned)__builtin_clz(t - 1);
        if (cp.windowLog > srcLog) cp.windowLog = srcLog;
    }
    if (cp.hashLog > cp.windowLog + 1) cp.hashLog = cp.windowLog + 1;
    if (cp.chainLog > cp.windowLog) cp.chainLog = cp.windowLog;
    if (cp.windowLog < 10) cp.windowLog = 10;
    *out = cp;
    return true;
}

// ZSTD_COMPRESSBOUND, lib/zstd.h:235
static inline size_t host_compress_bound(size_t n)
{
    return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

}  // namespace zhip
