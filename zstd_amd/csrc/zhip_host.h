// zhip_host.h — host-only helpers of libzstd_hip: parameter selection and bounds.
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace zhip {

struct CParams { unsigned windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy; };

// Level -> parameters for the strategies this library implements (fast 1, dfast 2, greedy 3, lazy 4, lazy2 5 — the
// last three with the hash-chain matcher, i.e. the reference with ZSTD_c_useRowMatchFinder = ZSTD_ps_disable).
// Values are the reference's rows (lib/compress/clevels.h:24-130: "base for negative levels", levels 1..12) for
// its four source-size classes; the selection + adjustment logic mirrors ZSTD_getCParams_internal
// (lib/compress/zstd_compress.c:7123-7145) and ZSTD_adjustCParams_internal (:1466-1602) for a known source size
// and no dictionary.  tests/test_host_params.py sweeps this against the real reference.
enum { HOST_CPM_NONE = 0, HOST_CPM_ATTACH = 1, HOST_CPM_CREATE_CDICT = 2 };       // ZSTD_cParamMode_e (zstd_compress_internal.h:176-191)
#define HOST_SRCSIZE_UNKNOWN (~0ULL)

// ZSTD_adjustCParams_internal (zstd_compress.c:1466-1602) for the strategies implemented here
static inline void host_adjust_cparams(CParams* cp, unsigned long long srcSize, unsigned long long dictSize, int mode)
{
    if (mode == HOST_CPM_CREATE_CDICT && dictSize && srcSize == HOST_SRCSIZE_UNKNOWN) srcSize = 513;   // :1524-1531
    if (mode == HOST_CPM_ATTACH) dictSize = 0;                                                           // :1532-1538
    if (srcSize <= (1ULL << 30) && dictSize <= (1ULL << 30)) {                                           // :1546-1553
        uint32_t const t = (uint32_t)(srcSize + dictSize);
        unsigned const srcLog = t < 64 ? 6 : 32 - (unsigned)__builtin_clz(t - 1);
        if (cp->windowLog > srcLog) cp->windowLog = srcLog;
    }
    if (srcSize != HOST_SRCSIZE_UNKNOWN) {                                                               // :1554-1560
        unsigned dawl = cp->windowLog;                                                                   // ZSTD_dictAndWindowLog :1432
        if (dictSize) {
            unsigned long long const windowSize = 1ULL << cp->windowLog;
            if (windowSize < dictSize + srcSize)
                dawl = (dictSize + windowSize >= (1ULL << 31)) ? 31 : 32 - (unsigned)__builtin_clz((uint32_t)(dictSize + windowSize) - 1);
        }
        if (cp->hashLog > dawl + 1) cp->hashLog = dawl + 1;
        if (cp->chainLog > dawl) cp->chainLog = dawl;
    }
    if (cp->windowLog < 10) cp->windowLog = 10;
    if (mode == HOST_CPM_CREATE_CDICT && cp->strategy <= 2) {                                            // :1568-1576 tagged tables
        if (cp->hashLog > 24) cp->hashLog = 24;
        if (cp->chainLog > 24) cp->chainLog = 24;
    }
}

// `overrides`: the explicitly set compression parameters of a CCtx (ZSTD_c_windowLog ... ZSTD_c_strategy), 0 = not set — they
// replace the level's row before the adjustment, exactly as ZSTD_getCParamsFromCCtxParams does (zstd_compress.c:1617-1644)
static inline bool host_get_cparams_mode(int level, unsigned long long srcSize, unsigned long long dictSize, int mode, CParams* out, const unsigned* overrides = nullptr)
{
    static const CParams rows[4][13] = {     // [size class][0 = negative-level base, 1..12 = level]; strategy 6+ = binary tree (not ours)
        /* srcSize > 256 KB */ {{19,12,13,1,6,1,1},{19,13,14,1,7,0,1},{20,15,16,1,6,0,1},{21,16,17,1,5,0,2},{21,18,18,1,5,0,2},
                                {21,18,19,3,5,2,3},{21,18,19,3,5,4,4},{21,19,20,4,5,8,4},{21,19,20,4,5,16,5},{22,20,21,4,5,16,5},
                                {22,21,22,5,5,16,5},{22,21,22,6,5,16,5},{22,22,23,6,5,32,5}},
        /* <= 256 KB       */ {{18,12,13,1,5,1,1},{18,13,14,1,6,0,1},{18,14,14,1,5,0,2},{18,16,16,1,4,0,2},{18,16,17,3,5,2,3},
                                {18,17,18,5,5,2,3},{18,18,19,3,5,4,4},{18,18,19,4,4,4,4},{18,18,19,4,4,8,5},{18,18,19,5,4,8,5},
                                {18,18,19,6,4,8,5},{18,18,19,5,4,12,6},{18,19,19,7,4,12,6}},
        /* <= 128 KB       */ {{17,12,12,1,5,1,1},{17,12,13,1,6,0,1},{17,13,15,1,5,0,1},{17,15,16,2,5,0,2},{17,17,17,2,4,0,2},
                                {17,16,17,3,4,2,3},{17,16,17,3,4,4,4},{17,16,17,3,4,8,5},{17,16,17,4,4,8,5},{17,16,17,5,4,8,5},
                                {17,16,17,6,4,8,5},{17,17,17,5,4,8,6},{17,18,17,7,4,12,6}},
        /* <= 16 KB        */ {{14,12,13,1,5,1,1},{14,14,15,1,5,0,1},{14,14,15,1,4,0,1},{14,14,15,2,4,0,2},{14,14,14,4,4,2,3},
                                {14,14,14,3,4,4,4},{14,14,14,4,4,8,5},{14,14,14,6,4,8,5},{14,14,14,8,4,8,5},{14,15,14,5,4,8,6},
                                {14,15,14,9,4,8,6},{14,15,14,3,4,12,7},{14,15,14,4,3,24,7}},
    };
    // ZSTD_getCParamRowSize (zstd_compress.c:7098-7116): attach mode ignores the dictionary; an unknown source with a
    // dictionary is taken as dictSize + 500 (the sum wraps exactly like the reference's U64 arithmetic)
    unsigned long long const rowDict = mode == HOST_CPM_ATTACH ? 0 : dictSize;
    bool const unknown = srcSize == HOST_SRCSIZE_UNKNOWN;
    unsigned long long const rSize = (unknown && rowDict == 0) ? HOST_SRCSIZE_UNKNOWN : srcSize + rowDict + ((unknown && rowDict > 0) ? 500 : 0);
    unsigned const cls = (rSize <= 256u * 1024) + (rSize <= 128u * 1024) + (rSize <= 16u * 1024);
    int row = level == 0 ? 3 : (level < 0 ? 0 : level);
    if (row > 12) return false;
    CParams cp = rows[cls][row];
    if (level < 0) { long const lv = level < -131072 ? -131072 : level; cp.targetLength = (unsigned)(-lv); }
    if (overrides) {
        unsigned* const f = &cp.windowLog;       // the seven fields in ZSTD_compressionParameters order
        for (int i = 0; i < 7; i++) if (overrides[i]) f[i] = overrides[i];
    }
    if (cp.strategy > 5) return false;       // btlazy2 and up: not implemented
    host_adjust_cparams(&cp, srcSize, dictSize, mode);
    *out = cp;
    return true;
}

static inline bool host_get_cparams(int level, unsigned long long srcSize, CParams* out, const unsigned* overrides = nullptr)
{
    return host_get_cparams_mode(level, srcSize, 0, HOST_CPM_NONE, out, overrides);
}

// ZSTD_checkCParams / ZSTD_cParam_getBounds (zstd_compress.c:433-470, lib/zstd.h:1236-1252) for explicitly set values (0 = not set)
static inline bool host_check_overrides(const unsigned ov[7])
{
    static const unsigned lo[7] = { 10, 6, 6, 1, 3, 0, 1 }, hi[7] = { 31, 30, 30, 30, 7, 131072, 9 };
    for (int i = 0; i < 7; i++) if (ov[i] && (ov[i] < lo[i] || ov[i] > hi[i])) return false;
    return true;
}

// ZSTD_COMPRESSBOUND, lib/zstd.h:235
// ZSTD_c_nbWorkers >= 1: how the reference's job pool cuts one frame (lib/compress/zstdmt_compress.c).  The result does not depend on
// the number of workers, only on the job size and the overlap.
static const size_t MT_JOBSIZE_MIN = (size_t)512 << 10;                 // ZSTDMT_JOBSIZE_MIN (zstdmt_compress.h:36); at or below it: no jobs
static inline size_t host_mt_job_size(const CParams& cp, unsigned long long jobSize)
{
    unsigned long long sec = jobSize;                                   // ZSTD_c_jobSize, clamped when set (zstdmt_compress.c:1250-1252)
    if (sec != 0 && sec < MT_JOBSIZE_MIN) sec = MT_JOBSIZE_MIN;
    if (sec > (1024ull << 20)) sec = 1024ull << 20;                     // ZSTDMT_JOBSIZE_MAX on 64-bit hosts
    if (sec == 0) { unsigned jl = cp.windowLog + 2; if (jl < 20) jl = 20; if (jl > 30) jl = 30; sec = 1ull << jl; }   // ZSTDMT_computeTargetJobLog (:1168-1180), no LDM
    return (size_t)sec;
}
static inline size_t host_mt_overlap_size(const CParams& cp, int overlapLog)
{
    // ZSTDMT_overlapLog_default (:1182-1203) by strategy, then ZSTDMT_computeOverlapSize (:1213-1233): 1 << (windowLog - (9 - overlapLog)), 0 at log 1
    int const dflt = cp.strategy == 9 ? 9 : (cp.strategy >= 7 ? 8 : (cp.strategy >= 5 ? 7 : 6));
    int const ov = overlapLog ? overlapLog : dflt;
    int const rlog = 9 - ov;
    int const ovLog = rlog >= 8 ? 0 : (int)cp.windowLog - rlog;
    return ovLog <= 0 ? 0 : (size_t)1 << ovLog;
}

static inline size_t host_compress_bound(size_t n)
{
    return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0);
}

}  // namespace zhip
