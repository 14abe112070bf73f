// zhip_lib.hip — host side of libzstd_hip.so (C ABI in include/zstd_hip.h) + kernel launches.  gfx950 only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <mutex>
#include <thread>
#include <chrono>
#include "../../include/zstd_hip.h"
#include "zhip_common.h"
// the kernels are compiled family by family (zhip_k_*.hip -> their own code objects); this translation unit only launches them:
// the family headers for their structures / LDS sizes / constants, then one prototype per kernel
#ifdef ZHIP_UNITY
#include "zhip_kernels.h"
#else
#include "zhip_parse.h"
#include "zhip_parse_dfast.h"
#include "zhip_parse_lazy.h"
#include "zhip_parse_dict.h"
#include "zhip_parse_ext.h"
#include "zhip_entropy.h"
#include "zhip_frame.h"
#include "zhip_frame_lazy.h"
#include "zhip_decode.h"
#include "zhip_decode_big.h"
#include "zhip_kernel_decls.h"
#endif
#include "zhip_host.h"
#include "zhip_cdict_host.h"

// zstd's error numbering (lib/zstd_errors.h:60-101): results are (size_t)-code
enum { ZE_GENERIC = 1, ZE_parameter_unsupported = 40, ZE_parameter_outOfBound = 42, ZE_memory_allocation = 64,
       ZE_dstSize_tooSmall = 70, ZE_srcSize_wrong = 72, ZE_sequenceProducer_failed = 106 };
#define ZERR(c) ((size_t)-(long)(c))

#define ZHIP_MAX_CHUNKS 8
// The ZSTD_fast stage's launch form (A/B in profiles/r03_ab_queue_forms.log): persistent wavefronts on a ticket
// queue, units dispatched by descending estimated cost, and four global-table wavefronts per CU beside the eight LDS-table ones (what the
// 141 / 145 registers of the two kernels leave room for).  $ZHIP_FAST_QUEUE=0 restores one workgroup per unit in index order.
#ifndef ZHIP_FAST_QUEUE_DEFAULT
#define ZHIP_FAST_QUEUE_DEFAULT 1
#endif
#ifndef ZHIP_FAST_ORDER_DEFAULT
#define ZHIP_FAST_ORDER_DEFAULT 1
#endif
#ifndef ZHIP_DICT_GWAVES_DEFAULT
#define ZHIP_DICT_GWAVES_DEFAULT 20      /* round 6: the queue kernels fit 72 registers = 28 wavefronts per CU, about eleven of them on LDS tables; 16 -> 20: -2.7 % (24: the same) */
#endif
#ifndef ZHIP_FAST_GWAVES_DEFAULT
#define ZHIP_FAST_GWAVES_DEFAULT 8      /* round 6: both kernels fit 128 registers = 16 wavefronts per CU, eight of them on the LDS tables */
#endif
#ifndef ZHIP_FAST_GWAVES_SPARSE
#define ZHIP_FAST_GWAVES_SPARSE 6       /* the global-table wavefronts per CU a batch of few sequences per unit gets (k_order_sort decides on the device) */
#endif
#ifndef ZHIP_FAST_DENSE_COST
#define ZHIP_FAST_DENSE_COST 4000u      /* mean k_order_cost estimate (sequences + bytes / 128) from which a batch counts as dense: datagen -P50 ~ 1 900, Silesia-shaped ~ 6 500, text ~ 11 900 */
#endif
#ifndef ZHIP_ENT_SMALL_PAD
#define ZHIP_ENT_SMALL_PAD 0          /* measurement only: extra dynamic LDS per record = fewer resident records per CU (occupancy slope) */
#endif
struct zhip_ctx_s {
    int device;
    size_t maxUnits;
    hipStream_t stream;
    hipEvent_t ev[5];
    // optional pipelining: the batch is cut into chunks whose match-finder / entropy kernels run on separate streams
    // (earlier chunks at higher priority), so a chunk's entropy stage overlaps the match finder of the later chunks
    int nChunks; hipStream_t cs[ZHIP_MAX_CHUNKS]; hipEvent_t cev[ZHIP_MAX_CHUNKS];
    // device scratch, one slot per unit
    ZhipUnit*  dUnits;
    ZhipSlot*  dSlots;                         // per unit: where its sequences / literals / output slot live
    ZhipSeq*   dSeqs;
    ZhipParse* dParse;
    uint8_t*   dLits;
    uint16_t*  dStBits;
    uint8_t*   dOut;
    uint32_t*  dTabs; size_t tabsCap;          // dfast / hash chain: per-unit tables in HBM, grown on demand
    size_t     tabStride;                      // words per unit of the current call (0 = strategy fast only)
    uint64_t*  dBest; size_t bestCap;          // hash chain: best[] records, ZHIP_UNIT_MAX per unit of a chunk
    size_t     hcChunk;                        // hash chain: units per pass over dTabs / dBest
    uint32_t   hcMaxLen, hcHashLog;            // hash chain: longest unit / largest hashLog of the call
    std::vector<hipEvent_t> hcEv; size_t hcEvUsed;   // hash chain: 4 events per chunk of the last call
    int checksum; uint32_t* dChecks;           // ZSTD_c_checksumFlag: per-unit XXH64 (low 32 bits), computed by k_xxh64 before the entropy stage
    const zhip::ZhipDictEntropy* curDictEntropy; uint32_t curDictID;   // dictionary entropy state of the current call (records path)
    int        strategy;                       // family mask of the current call's units: bit 0 fast, 1 dfast, 2 hash chain
    size_t     seqArena, litArena, outArena;   // arena capacities: sequences (entries), literal bytes, output-slot bytes
    uint32_t*  dOutSize;
    uint64_t*  dOutOff;
    // multi-block frames (zhip_frame.h): output room and per-frame chain state, grown on demand
    uint8_t* dFrameOut; size_t frameOutCap; zhip::ZhipFrameState* dFrameState; size_t frameStateCap;
    // ... as jobs (ZSTD_c_nbWorkers semantics): job table, whole-frame descriptors for the checksum, per-frame sizes
    zhip::ZhipJob* dJobs; size_t jobsCap; ZhipUnit* dFrameUnits; uint32_t* dFrameSizes; size_t frameUnitsCap;
    std::vector<zhip::ZhipJob> hJobs; std::vector<ZhipUnit> hFrameUnits; std::vector<uint32_t> hFrameSizes;
    // ... of the lazy strategies (zhip_frame_lazy.h): links, tags, records per position of every window; head tables; grown on demand
    uint32_t* dLzPrev = nullptr; uint8_t* dLzTags = nullptr; zhip::LzRec* dLzBest = nullptr; uint32_t* dLzHeads = nullptr; zhip::ZhipLzSlot* dLzSlots = nullptr;
    size_t lzPosCap = 0, lzHeadCap = 0, lzSlotCap = 0, lzRingCap = 0;
    uint8_t* dLzRing = nullptr; uint64_t lzRing = 0; int lzRingOn = 1;      // the row matcher's live rows (zhip_frame_lazy.h: LzRing); zhip_set_live_rows(0): live searches walk the links
    std::vector<zhip::ZhipLzSlot> hLz; bool lzAny = false, lzAll = false; uint64_t lzPos = 0, lzHeads = 0; uint32_t lzLongest = 0;
    // staging for the host-buffer API
    uint8_t* dSrcStage; size_t srcStageCap;
    uint8_t* dDstStage; size_t dstStageCap;
    // pinned host mirrors
    ZhipUnit* hUnits; uint32_t* hOutSize; ZhipParse* hParse; ZhipSlot* hSlots;
    // last call
    size_t nUnits; double timing[4]; unsigned long long stats[5];
    // sequence-producer cache
    int rowMode;                         // greedy / lazy / lazy2: 0 auto = the reference's default (row-hash matcher when windowLog > 14), 1 = ZSTD_ps_enable (the same, and units
                                         // with windowLog <= 14 are refused: the device has no row matcher for them), 2 = hash chain (ZSTD_ps_disable)
    uint32_t* dTileSums = nullptr; uint64_t* dTileOffs = nullptr; size_t scanCap = 0;    // tiles of the frame-size prefix sum (launch_offsets)
    bool wideFast = false;               // build_units met a ZSTD_fast unit with hashLog > 15
    // ZSTD_fast queue form (launch_parse): ticket counter, dispatch order + cost classes, the co-kernel's tables in global memory, its stream
    uint32_t* dQueue = nullptr; uint32_t* dOrder = nullptr; uint32_t* dCost = nullptr; uint32_t* dGTabs = nullptr; size_t gtabsCap = 0;
    hipStream_t coStream = nullptr; hipEvent_t coEv[2] = {nullptr, nullptr}; int numCUs = 0;
    int slotShare = 1;                   // the contexts that share this device's wavefront slots (a lane of zhip_compress_multi: the lanes of its device); launch_parse sizes its grids for 1 / slotShare of them
    int fastQueue = 0, fastOrder = 0, fastGWaves = 0;    // the ZSTD_fast stage: queue form, heaviest-first order, global-table wavefronts per CU (constants, A/B in profiles/r05_ab_fast_occupancy.log)
    int dfOccPerCU = 0;                                  // resident k_parse_dfast workgroups per CU (asked once)
    size_t fastOccSmem = ~(size_t)0; int fastOccPerCU = 1;
    int dictQueue = 1, dictGWaves = 0;                   // the records stage's queue form, global-table wavefronts per CU
    int rowDefault;                      // what mode 0 restores: the context's $ZHIP_ROW_MATCHER default, captured at creation
    int rhPredict = 0, lzPredict = 1;    // the row matcher's two-pass prediction, units / frames (zhip_set_prediction).  Units: OFF — a live search reads the row's list (rh_live_lists),
                                         // one parse is faster.  Frames: ON, behind a 32 KB probe per window (1 MiB datagen frames 681 -> 366 ms; text frames 168 -> 171 ms with the
                                         // probe, 224 without; profiles/r04_live_rows.log, r04_predict_probe.log)
    unsigned ovr[7]; bool haveOvr;       // explicit compression parameters of the call in progress (zhip_compress_params*), 0 = level's own
    const void* cacheSrc; size_t cacheSize, cacheBlock; int cacheLevel;
    std::vector<uint64_t> cacheHash;     // two 64-bit content hashes per prepared block: a hit must also match the bytes
    std::vector<ZhipSeq> cacheSeqs; std::vector<ZhipParse> cacheParse; std::vector<ZhipUnit> cacheUnits;
    std::vector<uint64_t> cacheSeqOff;   // where each prepared block's records start in cacheSeqs
    ZhipSeq* dSeqPack; size_t seqPackCap; uint64_t* dSeqPackOff; size_t seqPackOffCap;   // device side of the packed copy (k_seq_compact)
    std::mutex cacheMu;                  // guards the prepared-block cache alone: producer callbacks of many host threads only share this one
    std::mutex mu;
    char err[256];
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    return ZERR(ZE_GENERIC); } } while (0)

extern "C" {

int zhip_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

unsigned zhip_isError(size_t code) { return code > ZERR(120); }
const char* zhip_getErrorName(size_t code)
{
    if (!zhip_isError(code)) return "No error detected";
    switch ((int)(0 - code)) {
    case ZE_GENERIC: return "Error (generic)";
    case ZE_parameter_unsupported: return "Unsupported parameter";
    case ZE_parameter_outOfBound: return "Parameter is out of bound";
    case ZE_memory_allocation: return "Allocation error : not enough memory";
    case ZE_dstSize_tooSmall: return "Destination buffer is too small";
    case ZE_srcSize_wrong: return "Src size is incorrect";
    case ZE_sequenceProducer_failed: return "Block-level external sequence producer returned an error code";
    default: return "Unspecified error code";
    }
}

size_t zhip_compressBound(size_t srcSize, size_t unitSize)
{
    if (unitSize == 0 || unitSize > ZHIP_UNIT_MAX) unitSize = ZHIP_UNIT_MAX;
    size_t const full = srcSize / unitSize, tail = srcSize % unitSize;
    size_t b = full * zhip::host_compress_bound(unitSize);
    if (tail || srcSize == 0) b += zhip::host_compress_bound(tail);
    return b;
}

int zhip_getCParams(int level, unsigned long long srcSize, unsigned out[7])
{
    zhip::CParams cp;
    if (!zhip::host_get_cparams(level, srcSize, &cp)) return -1;
    out[0] = cp.windowLog; out[1] = cp.chainLog; out[2] = cp.hashLog; out[3] = cp.searchLog;
    out[4] = cp.minMatch; out[5] = cp.targetLength; out[6] = cp.strategy;
    return 0;
}

const char* zhip_last_error(const zhip_ctx* ctx) { return ctx ? ctx->err : "null context"; }

void zhip_destroy(zhip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->coStream) (void)hipStreamSynchronize(c->coStream);      // the global-table co-kernels use dGTabs / dQueue / dOrder (a call that returned early on an error may have left one running)
    (void)hipFree(c->dUnits); (void)hipFree(c->dSlots); (void)hipFree(c->dSeqs); (void)hipFree(c->dParse); (void)hipFree(c->dLits); (void)hipFree(c->dStBits);
    (void)hipFree(c->dOut); (void)hipFree(c->dOutSize); (void)hipFree(c->dOutOff); (void)hipFree(c->dTabs); (void)hipFree(c->dBest); (void)hipFree(c->dChecks); (void)hipFree(c->dTileSums); (void)hipFree(c->dTileOffs);
    (void)hipFree(c->dSeqPack); (void)hipFree(c->dSeqPackOff);
    (void)hipFree(c->dSrcStage); (void)hipFree(c->dDstStage); (void)hipFree(c->dFrameOut); (void)hipFree(c->dFrameState);
    (void)hipFree(c->dJobs); (void)hipFree(c->dFrameUnits); (void)hipFree(c->dFrameSizes);
    (void)hipFree(c->dLzPrev); (void)hipFree(c->dLzTags); (void)hipFree(c->dLzBest); (void)hipFree(c->dLzHeads); (void)hipFree(c->dLzSlots); (void)hipFree(c->dLzRing);
    (void)hipFree(c->dQueue); (void)hipFree(c->dOrder); (void)hipFree(c->dCost); (void)hipFree(c->dGTabs);
    if (c->coStream) (void)hipStreamDestroy(c->coStream);
    for (int i = 0; i < 2; i++) if (c->coEv[i]) (void)hipEventDestroy(c->coEv[i]);
    (void)hipHostFree(c->hUnits); (void)hipHostFree(c->hOutSize); (void)hipHostFree(c->hParse); (void)hipHostFree(c->hSlots);
    for (int i = 0; i < 5; i++) (void)hipEventDestroy(c->ev[i]);
    for (hipEvent_t e : c->hcEv) (void)hipEventDestroy(e);
    for (int i = 0; i < ZHIP_MAX_CHUNKS; i++) { if (c->cs[i]) (void)hipStreamDestroy(c->cs[i]); if (c->cev[i]) (void)hipEventDestroy(c->cev[i]); }
    (void)hipStreamDestroy(c->stream);
    delete c;
}

static zhip_ctx* create_impl(int device, size_t maxUnits, size_t seqArena, size_t litArena, size_t outArena)
{
    if (maxUnits == 0) maxUnits = 1;
    {   int count = 0;                                  // (a failed hipSetDevice would leave a sticky error for the next launch check of another context)
        if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) { (void)hipGetLastError(); return nullptr; }
    }
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    zhip_ctx* c = new zhip_ctx_s();
    c->dUnits = nullptr; c->dSlots = nullptr; c->hSlots = nullptr; c->dSeqs = nullptr; c->dParse = nullptr; c->dLits = nullptr; c->dStBits = nullptr; c->dOut = nullptr;
    c->dOutSize = nullptr; c->dOutOff = nullptr; c->hUnits = nullptr; c->hOutSize = nullptr; c->hParse = nullptr;
    c->dTabs = nullptr; c->tabsCap = 0; c->tabStride = 0; c->strategy = 1; c->dBest = nullptr; c->bestCap = 0; c->hcChunk = 0; c->hcMaxLen = 0; c->hcEvUsed = 0; c->curDictEntropy = nullptr; c->curDictID = 0; c->checksum = 0; c->dChecks = nullptr;
    c->seqArena = seqArena; c->litArena = litArena; c->outArena = outArena;
    c->device = device; c->maxUnits = maxUnits; c->err[0] = 0; c->nUnits = 0;
    c->cacheSrc = nullptr; c->cacheSize = 0; c->cacheBlock = 0; c->cacheLevel = 0;
    c->dSeqPack = nullptr; c->seqPackCap = 0; c->dSeqPackOff = nullptr; c->seqPackOffCap = 0;
    memset(c->ovr, 0, sizeof(c->ovr)); c->haveOvr = false;
    {   const char* e = getenv("ZHIP_ROW_MATCHER"); c->rowMode = c->rowDefault = (e && (!strcmp(e, "disable") || !strcmp(e, "0"))) ? 2 : 0; }
    c->lzRingOn = 1; c->rhPredict = 0; c->lzPredict = 1;      // live rows on; prediction: units off (one parse is faster with the rows, profiles/r04_live_rows.log), frames on behind the 32 KB probe (zhip_set_live_rows / zhip_set_prediction)
    c->dSrcStage = nullptr; c->srcStageCap = 0; c->dDstStage = nullptr; c->dstStageCap = 0;
    c->dFrameOut = nullptr; c->frameOutCap = 0; c->dFrameState = nullptr; c->frameStateCap = 0;
    c->dJobs = nullptr; c->jobsCap = 0; c->dFrameUnits = nullptr; c->dFrameSizes = nullptr; c->frameUnitsCap = 0;
    for (int i = 0; i < ZHIP_MAX_CHUNKS; i++) { c->cs[i] = nullptr; c->cev[i] = nullptr; }
    memset(c->timing, 0, sizeof(c->timing));
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 5 && ok; i++) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    {   const char* e = getenv("ZHIP_PIPELINE_CHUNKS");
        c->nChunks = e ? atoi(e) : 1;
        if (c->nChunks < 1) c->nChunks = 1;
        if (c->nChunks > ZHIP_MAX_CHUNKS) c->nChunks = ZHIP_MAX_CHUNKS;
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);      // hi = numerically lowest = highest priority
        for (int i = 0; i < c->nChunks && ok && c->nChunks > 1; i++) {
            int pr = hi + i; if (pr > lo) pr = lo;
            ok = hipStreamCreateWithPriority(&c->cs[i], hipStreamNonBlocking, pr) == hipSuccess;
            ok = ok && hipEventCreateWithFlags(&c->cev[i], hipEventDisableTiming) == hipSuccess;
        }
    }
    ok = ok && hipMalloc((void**)&c->dUnits, maxUnits * sizeof(ZhipUnit)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dSlots, maxUnits * sizeof(ZhipSlot)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->hSlots, maxUnits * sizeof(ZhipSlot), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dSeqs, seqArena * sizeof(ZhipSeq)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dParse, maxUnits * sizeof(ZhipParse)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dLits, litArena) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dStBits, seqArena * 3 * sizeof(uint16_t)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dOut, outArena) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dOutSize, (maxUnits + 1) * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dOutOff, (maxUnits + 1) * sizeof(uint64_t)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->hUnits, maxUnits * sizeof(ZhipUnit), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->hOutSize, (maxUnits + 1) * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->hParse, maxUnits * sizeof(ZhipParse), hipHostMallocDefault) == hipSuccess;
    {   // the queue form of the ZSTD_fast stage (launch_parse)
        c->fastQueue = ZHIP_FAST_QUEUE_DEFAULT; c->fastOrder = ZHIP_FAST_ORDER_DEFAULT; c->fastGWaves = ZHIP_FAST_GWAVES_DEFAULT;
        c->dictQueue = 1; c->dictGWaves = ZHIP_DICT_GWAVES_DEFAULT;
        hipDeviceProp_t prop;
        c->numCUs = (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        ok = ok && hipMalloc((void**)&c->dQueue, 64) == hipSuccess;
        ok = ok && hipMalloc((void**)&c->dOrder, maxUnits * sizeof(uint32_t)) == hipSuccess;
        ok = ok && hipMalloc((void**)&c->dCost, maxUnits * sizeof(uint32_t)) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&c->coStream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; i < 2 && ok; i++) ok = hipEventCreateWithFlags(&c->coEv[i], hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { zhip_destroy(c); return nullptr; }
    return c;
}

zhip_ctx* zhip_create(int device, size_t maxUnits)
{
    if (maxUnits == 0) maxUnits = 1;
    return create_impl(device, maxUnits, maxUnits * (size_t)ZHIP_SEQ_CAP, maxUnits * (size_t)ZHIP_LIT_STRIDE, maxUnits * (size_t)ZHIP_OUT_STRIDE);
}

// per-record slot sizes of the packed layout (records path)
static inline size_t rec_seq_cap(size_t n) { return n / 4 + 8; }
static inline size_t rec_lit_bytes(size_t n) { return (n + 64 + 15) & ~(size_t)15; }
static inline size_t rec_out_bytes(size_t n) { return (zhip::host_compress_bound(n) + 32 + 15) & ~(size_t)15; }

zhip_ctx* zhip_create_for_records(int device, size_t maxRecords, size_t maxTotalBytes)
{
    if (maxRecords == 0) maxRecords = 1;
    return create_impl(device, maxRecords, maxTotalBytes / 4 + 8 * maxRecords + 64, maxTotalBytes + 80 * maxRecords + 64,
                       maxTotalBytes + (maxTotalBytes >> 8) + 128 * maxRecords + 64);
}

int zhip_set_row_matcher(zhip_ctx* c, int mode)
{
    std::lock_guard<std::mutex> lk(c->mu);
    if (mode < -1 || mode > 2) return 1;
    // -1: back to the context's own default ($ZHIP_ROW_MATCHER at creation); 0 = ZSTD_ps_auto: the reference's default; 1 = ZSTD_ps_enable: the reference then uses the row matcher whatever the window
    // (ZSTD_resolveRowMatchFinderMode returns an explicit mode unchanged, zstd_compress.c:244) — units it would give windowLog <= 14 are
    // refused when they are planned rather than compressed differently; 2 = ZSTD_ps_disable
    c->rowMode = mode < 0 ? c->rowDefault : mode;
    return 0;
}

// the row matcher's two-pass prediction (zhip_parse_lazy.h: rh_reconcile; zhip_frame_lazy.h: frame_lazy_predict): 1 on, 0 off, -1 unchanged.
// Same bytes either way — it only changes how many searches the exact parse has to redo live.
int zhip_set_prediction(zhip_ctx* c, int units, int frames)
{
    std::lock_guard<std::mutex> lk(c->mu);
    if (units < -1 || units > 1 || frames < -1 || frames > 1) return 1;
    if (units >= 0) c->rhPredict = units;
    if (frames >= 0) c->lzPredict = frames;
    return 0;
}

// the row matcher's live rows of the FRAME kernels (LzRing, zhip_frame_lazy.h): 1 on (default), 0 = their live searches walk the links instead — the form a
// context also falls back to when the rows' arena cannot be allocated.  Same bytes either way.  (The unit kernels keep no rows since round 5: a live
// search there reads the row's list, zhip_parse_lazy.h: rh_live_lists.)
int zhip_set_live_rows(zhip_ctx* c, int on)
{
    std::lock_guard<std::mutex> lk(c->mu);
    c->lzRingOn = on ? 1 : 0;
    return 0;
}

int zhip_set_frame_checksum(zhip_ctx* c, int enable)
{
    std::lock_guard<std::mutex> lk(c->mu);
    c->checksum = enable ? 1 : 0;
    return 0;
}

void zhip_last_timing(const zhip_ctx* c, double t[4]) { for (int i = 0; i < 4; i++) t[i] = c->timing[i]; }

void zhip_last_hc_timing(zhip_ctx* c, double t[3])
{
    std::lock_guard<std::mutex> lk(c->mu);
    t[0] = t[1] = t[2] = 0;
    for (size_t i = 0; i + 4 <= c->hcEvUsed; i += 4)
        for (int k = 0; k < 3; k++) { float ms = 0; if (hipEventElapsedTime(&ms, c->hcEv[i + k], c->hcEv[i + k + 1]) == hipSuccess) t[k] += ms; }
}

// ---- stage-test hooks: the wave-wide entropy-table builders on caller-supplied histograms (host buffers in and out)
static size_t test_copy_back(zhip_ctx* c, void* h, const void* d, size_t n) { HIPCHK(c, hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); return 0; }
size_t zhip_test_huf_tables(zhip_ctx* c, const unsigned* counts, const unsigned* maxSyms, unsigned nCases, unsigned maxNbBits,
                            unsigned* codes, unsigned char* hdrs, unsigned* meta)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t *dC = nullptr, *dM = nullptr, *dCode = nullptr, *dMeta = nullptr; uint8_t* dH = nullptr;
    size_t const n = nCases;
    HIPCHK(c, hipMalloc((void**)&dC, n * 1024)); HIPCHK(c, hipMalloc((void**)&dM, n * 4 + 4)); HIPCHK(c, hipMalloc((void**)&dCode, n * 1024));
    HIPCHK(c, hipMalloc((void**)&dH, n * 136 + 4)); HIPCHK(c, hipMalloc((void**)&dMeta, n * 8 + 4));
    HIPCHK(c, hipMemcpy(dC, counts, n * 1024, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(dM, maxSyms, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(zhip::k_test_huf, dim3(nCases), dim3(64), sizeof(zhip::ZhipTestHufShared), c->stream, dC, dM, maxNbBits, dCode, dH, dMeta);
    HIPCHK(c, hipGetLastError()); HIPCHK(c, hipStreamSynchronize(c->stream));
    size_t e = test_copy_back(c, codes, dCode, n * 1024); if (!e) e = test_copy_back(c, hdrs, dH, n * 136); if (!e) e = test_copy_back(c, meta, dMeta, n * 8);
    (void)hipFree(dC); (void)hipFree(dM); (void)hipFree(dCode); (void)hipFree(dH); (void)hipFree(dMeta);
    return e;
}
size_t zhip_test_fse_tables(zhip_ctx* c, const unsigned* counts, const unsigned* params, unsigned nCases,
                            short* norms, unsigned char* ncounts, int* meta, void* tables, size_t tableStride)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (tableStride != sizeof(zhip::FseCTable)) return ZERR(ZE_GENERIC);
    uint32_t *dC = nullptr, *dP = nullptr; int16_t* dN = nullptr; uint8_t* dH = nullptr; int32_t* dMeta = nullptr; zhip::FseCTable* dT = nullptr;
    size_t const n = nCases;
    HIPCHK(c, hipMalloc((void**)&dC, n * 256)); HIPCHK(c, hipMalloc((void**)&dP, n * 16)); HIPCHK(c, hipMalloc((void**)&dN, n * 128));
    HIPCHK(c, hipMalloc((void**)&dH, n * 64)); HIPCHK(c, hipMalloc((void**)&dMeta, n * 8)); HIPCHK(c, hipMalloc((void**)&dT, n * sizeof(zhip::FseCTable)));
    HIPCHK(c, hipMemcpy(dC, counts, n * 256, hipMemcpyHostToDevice)); HIPCHK(c, hipMemcpy(dP, params, n * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(zhip::k_test_fse, dim3(nCases), dim3(64), sizeof(zhip::ZhipTestFseShared), c->stream, dC, dP, dN, dH, dMeta, dT);
    HIPCHK(c, hipGetLastError()); HIPCHK(c, hipStreamSynchronize(c->stream));
    size_t e = test_copy_back(c, norms, dN, n * 128); if (!e) e = test_copy_back(c, ncounts, dH, n * 64); if (!e) e = test_copy_back(c, meta, dMeta, n * 8);
    if (!e) e = test_copy_back(c, tables, dT, n * sizeof(zhip::FseCTable));
    (void)hipFree(dC); (void)hipFree(dP); (void)hipFree(dN); (void)hipFree(dH); (void)hipFree(dMeta); (void)hipFree(dT);
    return e;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ internals
// fill ctx->hUnits for `srcSize` bytes cut into unitSize chunks; returns number of units or 0 with *err set

static size_t build_units(zhip_ctx* c, size_t srcSize, size_t unitSize, int level, size_t* err, uint32_t* maxHashLog)
{
    if (unitSize == 0 || unitSize > ZHIP_UNIT_MAX) { *err = ZERR(ZE_parameter_outOfBound); return 0; }
    size_t const nUnits = srcSize ? (srcSize + unitSize - 1) / unitSize : 1;
    if (nUnits > c->maxUnits) { snprintf(c->err, sizeof(c->err), "%zu units > context capacity %zu", nUnits, c->maxUnits); *err = ZERR(ZE_srcSize_wrong); return 0; }
    zhip::CParams full, tail; bool haveFull = false;
    uint32_t mh = 0; size_t tabWords = 0; int fam = 0; uint32_t hcMaxLen = 0, hcHlog = 6;
    for (size_t i = 0; i < nUnits; i++) {
        size_t const off = i * unitSize;
        size_t const len = srcSize - off < unitSize ? srcSize - off : unitSize;
        zhip::CParams* cp = &tail;
        const unsigned* const ov = c->haveOvr ? c->ovr : nullptr;
        if (len == unitSize) { if (!haveFull) { if (!zhip::host_get_cparams(level, len, &full, ov)) { *err = ZERR(ZE_parameter_unsupported); return 0; } haveFull = true; } cp = &full; }
        else if (!zhip::host_get_cparams(level, len, &tail, ov)) { *err = ZERR(ZE_parameter_unsupported); return 0; }
        if (len > 1 && ((size_t)1 << cp->windowLog) < len) {       // an explicit window smaller than the unit: matches would have to respect it
            snprintf(c->err, sizeof(c->err), "windowLog %u is smaller than a %zu-byte unit: not implemented on device", cp->windowLog, len); *err = ZERR(ZE_parameter_unsupported); return 0; }
        ZhipUnit& u = c->hUnits[i];
        u.srcOff = off; u.srcLen = (uint32_t)len;
        {   ZhipSlot& sl = c->hSlots[i]; sl.seqOff = i * (uint64_t)ZHIP_SEQ_CAP; sl.litOff = i * (uint64_t)ZHIP_LIT_STRIDE; sl.outOff = i * (uint64_t)ZHIP_OUT_STRIDE; sl.seqCap = ZHIP_SEQ_CAP; sl.pad0 = 0; }
        u.windowLog = (uint8_t)cp->windowLog; u.chainLog = (uint8_t)cp->chainLog; u.hashLog = (uint8_t)cp->hashLog;
        u.minMatch = (uint8_t)cp->minMatch; u.strategy = (uint8_t)cp->strategy; u.searchLog = (uint8_t)cp->searchLog;
        u.litMode = (cp->strategy == ZHIP_STRAT_FAST && cp->targetLength > 0) ? 1 : 0;
        u.pad0 = 0;
        u.targetLength = cp->targetLength;
        // ZSTD_resolveRowMatchFinderMode (zstd_compress.c:237-253): greedy / lazy / lazy2 use the row-hash matcher when windowLog > 14
        u.rowLog = 0; u.pad1 = 0;
        if (cp->strategy >= ZHIP_STRAT_GREEDY && cp->strategy <= ZHIP_STRAT_LAZY2 && c->rowMode != 2 && cp->windowLog > 14)
            u.rowLog = cp->searchLog < 4 ? 4 : (cp->searchLog > 6 ? 6 : cp->searchLog);       // :2042 BOUNDED(4, searchLog, 6)
        else if (cp->strategy >= ZHIP_STRAT_GREEDY && cp->strategy <= ZHIP_STRAT_LAZY2 && c->rowMode == 1) {
            snprintf(c->err, sizeof(c->err), "ZSTD_ps_enable with windowLog %u <= 14: the row matcher for such windows is not implemented on device", cp->windowLog);
            *err = ZERR(ZE_parameter_unsupported); return 0;
        }
        // a call may mix families (a ragged tail takes the row of its own size class, e.g. level 4: dfast + greedy tail)
        if (cp->strategy == ZHIP_STRAT_FAST) { fam |= 1; if (cp->hashLog > mh) mh = cp->hashLog; }
        else if (cp->strategy == ZHIP_STRAT_DFAST) { fam |= 2; size_t const w = zhip::dfast_table_bytes(cp->hashLog, cp->chainLog) >> 2; if (w > tabWords) tabWords = w; }
        else if (cp->strategy <= ZHIP_STRAT_LAZY2) { fam |= 4; size_t const w = zhip::hc_table_words(cp->hashLog); if (w > tabWords) tabWords = w; if (len > hcMaxLen) hcMaxLen = (uint32_t)len; if (cp->hashLog > hcHlog) hcHlog = cp->hashLog; }
        else { snprintf(c->err, sizeof(c->err), "strategy %u not implemented on device", cp->strategy); *err = ZERR(ZE_parameter_unsupported); return 0; }
    }
    if (mh > 15) {      // the unit kernel's LDS table ends at 2^15 entries: the caller sends such units through the frame kernel, whose table policy reaches HBM
        c->wideFast = true; snprintf(c->err, sizeof(c->err), "hashLog %u does not fit the unit kernel's LDS table", mh); *err = ZERR(ZE_parameter_unsupported); return 0; }
    c->strategy = fam ? fam : 1; c->tabStride = (tabWords + 3) & ~(size_t)3; c->hcMaxLen = hcMaxLen; c->hcHashLog = hcHlog;
    {   // hash chain: dTabs / dBest hold one chunk of units at a time (1 MB + 1 MB per 128 KB unit)
        size_t const chunk = 8192;                     // level 5, 1 GiB: 8192 units per chunk 185 ms, 4096: 192 ms, 2048: 263 ms (profiles/r05_ab_l5_rowlists.log) — 21 GB of tables, records and live rows at 8192
        c->hcChunk = nUnits < chunk ? nUnits : chunk;
    }
    // dfast: one table pair per RESIDENT workgroup (k_parse_dfast's workgroups are persistent and reuse theirs: at most 32 wavefronts per CU), not per unit
    size_t const dfPairs = nUnits < (size_t)32 * (size_t)c->numCUs ? nUnits : (size_t)32 * (size_t)c->numCUs;
    if (fam & 4) {
        // hash chain / row matcher: links + lists and the match records of one CHUNK of units.  When the device cannot give a chunk's worth (8 192 units of
        // 128 KB: 8 GB + 8 GB) the chunk is halved until it fits — the launch loop takes any chunk size (round-5 advisor finding: a failed hipMalloc used to
        // end the call with no smaller retry).  Buffers that are already large enough are kept.
        for (;;) {
            bool ok = true;
            if (c->tabsCap < c->hcChunk * c->tabStride) {
                (void)hipFree(c->dTabs); c->dTabs = nullptr; c->tabsCap = 0;
                if (hipMalloc((void**)&c->dTabs, c->hcChunk * c->tabStride * sizeof(uint32_t)) == hipSuccess) c->tabsCap = c->hcChunk * c->tabStride;
                else { (void)hipGetLastError(); ok = false; }
            }
            if (ok && c->bestCap < c->hcChunk * (size_t)ZHIP_UNIT_MAX) {
                (void)hipFree(c->dBest); c->dBest = nullptr; c->bestCap = 0;
                if (hipMalloc((void**)&c->dBest, c->hcChunk * (size_t)ZHIP_UNIT_MAX * sizeof(uint64_t)) == hipSuccess) c->bestCap = c->hcChunk * (size_t)ZHIP_UNIT_MAX;
                else { (void)hipGetLastError(); ok = false; }
            }
            if (ok) break;
            if (c->hcChunk <= 1) {
                snprintf(c->err, sizeof(c->err), "cannot allocate the match finder's tables and records for even one unit (%zu + %zu bytes)",
                         c->tabStride * sizeof(uint32_t), (size_t)ZHIP_UNIT_MAX * sizeof(uint64_t));
                *err = ZERR(ZE_memory_allocation); return 0;
            }
            c->hcChunk = (c->hcChunk + 1) / 2;
        }
    } else if ((fam & 2) && c->tabsCap < dfPairs * c->tabStride) {
        (void)hipFree(c->dTabs); c->dTabs = nullptr; c->tabsCap = 0;
        if (hipMalloc((void**)&c->dTabs, dfPairs * c->tabStride * sizeof(uint32_t)) != hipSuccess) { snprintf(c->err, sizeof(c->err), "cannot allocate %zu bytes of match-finder tables", dfPairs * c->tabStride * sizeof(uint32_t)); *err = ZERR(ZE_memory_allocation); return 0; }
        c->tabsCap = dfPairs * c->tabStride;
    }
    if ((fam & 2) && (fam & 4)) {                  // a mixed batch: room for the dfast workgroups' pairs as well
        if (c->tabsCap < dfPairs * c->tabStride) {
            (void)hipFree(c->dTabs); c->dTabs = nullptr; c->tabsCap = 0;
            if (hipMalloc((void**)&c->dTabs, dfPairs * c->tabStride * sizeof(uint32_t)) != hipSuccess) { *err = ZERR(ZE_memory_allocation); return 0; }
            c->tabsCap = dfPairs * c->tabStride;
        }
    }
    *maxHashLog = mh;
    return nUnits;
}

// exclusive prefix sum of the frame sizes: one workgroup up to 64 K units, tiles above (10 M records: 37 ms -> well under one)
static size_t launch_offsets(zhip_ctx* c, size_t nUnits, hipStream_t s)
{
    if (nUnits <= 65536) {
        hipLaunchKernelGGL(zhip::k_offsets, dim3(1), dim3(256), 0, s, c->dOutSize, (uint32_t)nUnits, c->dOutOff);
        return 0;
    }
    size_t const nTiles = (nUnits + ZHIP_SCAN_TILE - 1) / ZHIP_SCAN_TILE;
    if (c->scanCap < nTiles) {
        (void)hipFree(c->dTileSums); (void)hipFree(c->dTileOffs); c->dTileSums = nullptr; c->dTileOffs = nullptr; c->scanCap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dTileSums, nTiles * sizeof(uint32_t)));
        HIPCHK(c, hipMalloc((void**)&c->dTileOffs, (nTiles + 1) * sizeof(uint64_t)));
        c->scanCap = nTiles;
    }
    hipLaunchKernelGGL(zhip::k_offsets_tiles, dim3((unsigned)nTiles), dim3(256), 0, s, c->dOutSize, (uint32_t)nUnits, c->dTileSums);
    hipLaunchKernelGGL(zhip::k_offsets, dim3(1), dim3(256), 0, s, c->dTileSums, (uint32_t)nTiles, c->dTileOffs);
    hipLaunchKernelGGL(zhip::k_offsets_apply, dim3((unsigned)nTiles), dim3(256), 0, s, c->dOutSize, (uint32_t)nUnits, c->dTileOffs, (uint32_t)nTiles, c->dOutOff);
    return 0;
}

static size_t launch_parse(zhip_ctx* c, const uint8_t* srcDev, size_t nUnits, uint32_t maxHashLog, hipStream_t s)
{
    size_t smem = zhip::fast_tag_lds_bytes(maxHashLog);
    HIPCHK(c, hipMemcpyAsync(c->dUnits, c->hUnits, nUnits * sizeof(ZhipUnit), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->dSlots, c->hSlots, nUnits * sizeof(ZhipSlot), hipMemcpyHostToDevice, s));
    if (smem > 64 * 1024 && (c->strategy & 1))
        HIPCHK(c, hipFuncSetAttribute((const void*)zhip::k_parse_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    // one launch per strategy family present; every kernel skips the units of the other families
    if (c->strategy & 2)
    {
        size_t const ldsB = zhip::dfast_lds_bytes();
        if (ldsB > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)zhip::k_parse_dfast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB));
        // persistent workgroups — as many as are resident — drawing units from a ticket counter: one table pair per RESIDENT wavefront
        // (A/B against one workgroup and one table pair per unit, profiles/r04_ab_dfast_persistent.log: the same time — the kernel is latency-bound either way —
        // with 1.5 GB of tables instead of 40 GB on Silesia-shaped x64)
        unsigned grid = (unsigned)nUnits; uint32_t* dfQueue = nullptr;
        if (nUnits > 1) {
            if (!c->dfOccPerCU) {
                int perCU_ = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU_, (const void*)zhip::k_parse_dfast, 64, ldsB) != hipSuccess || perCU_ < 1) { (void)hipGetLastError(); perCU_ = 16; }
                c->dfOccPerCU = perCU_;
            }
            size_t resident = (size_t)c->dfOccPerCU * (size_t)c->numCUs;
            if (c->tabStride && resident > c->tabsCap / c->tabStride) resident = c->tabsCap / c->tabStride;       // never more workgroups than table pairs
            if (resident < nUnits) { grid = (unsigned)resident; dfQueue = c->dQueue + 4; HIPCHK(c, hipMemsetAsync(dfQueue, 0, 4, s)); }
        }
        hipLaunchKernelGGL(zhip::k_parse_dfast, dim3(grid), dim3(64), ldsB, s,
                           srcDev, c->dUnits, c->dSlots, (uint32_t)nUnits, c->dTabs, c->tabStride, c->dSeqs, c->dLits, c->dParse, dfQueue);
    }
    if ((c->strategy & 1) && c->fastQueue && nUnits > 1) {
        // queue form: persistent wavefronts draw units from one ticket counter — the LDS-table kernel on this stream and, beside it on
        // the same CUs, the global-table kernel on coStream (its wavefronts need no LDS); heaviest units first when an order is asked for
        if (smem > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)zhip::k_parse_fast_q, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (c->fastOccSmem != smem) {                                         // resident LDS-form workgroups per CU for this table size (nine at hashLog 13)
            int perCU_ = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU_, (const void*)zhip::k_parse_fast_q, 64, smem) != hipSuccess || perCU_ < 1) { (void)hipGetLastError(); perCU_ = 1; }
            c->fastOccSmem = smem; c->fastOccPerCU = perCU_;
        }
        int const perCU = c->fastOccPerCU;
        // a lane of the host-buffer path shares the device with the other lanes' chunks: each takes its share of BOTH kinds of slots, so that a chunk larger than
        // its share of the LDS slots spills to the global-table form instead of queueing behind the other lane's wavefronts (slotShare = 1: the whole device)
        size_t const share = c->slotShare > 1 ? (size_t)c->slotShare : 1;
        size_t const slotsQ = ((size_t)perCU * (size_t)c->numCUs + share - 1) / share;
        size_t const gridQ = slotsQ < nUnits ? slotsQ : nUnits;
        size_t gridG = ((size_t)c->fastGWaves * (size_t)c->numCUs + share - 1) / share;
        if (gridQ >= nUnits) gridG = 0;                                       // everything is resident on the LDS form already
        else if (gridG > nUnits - gridQ) gridG = nUnits - gridQ;
        bool const wantOrder = c->fastOrder && nUnits > 2;
        const uint32_t* order = nullptr;
        HIPCHK(c, hipMemsetAsync(c->dQueue, 0, 64, s));                      // the ticket counter
        if (wantOrder) {
            if (c->fastOrder == 2) hipLaunchKernelGGL(zhip::k_order_cost_stale, dim3((unsigned)((nUnits + 255) / 256)), dim3(256), 0, s, c->dParse, (uint32_t)nUnits, c->dCost);
            else hipLaunchKernelGGL(zhip::k_order_cost, dim3((unsigned)nUnits), dim3(64), 0, s, srcDev, c->dUnits, (uint32_t)nUnits, c->dCost);
            // (the sort also sets the queue's second word: how many of the global-table workgroups take part, by the batch's mean cost)
            size_t const gSparse = ((size_t)ZHIP_FAST_GWAVES_SPARSE * (size_t)c->numCUs + share - 1) / share;
            bool const decide = c->fastOrder == 1 && c->fastGWaves > ZHIP_FAST_GWAVES_SPARSE && gridG > gSparse;
            hipLaunchKernelGGL(zhip::k_order_sort, dim3(1), dim3(1024), 0, s, c->dCost, (uint32_t)nUnits, c->dOrder,
                               decide ? c->dQueue + 1 : (uint32_t*)nullptr, (uint32_t)gSparse, ZHIP_FAST_DENSE_COST);
            order = c->dOrder;
        }
        uint32_t const gtabWords = 1u << maxHashLog;
        if (gridG) {
            if (c->gtabsCap < gridG * gtabWords) {
                (void)hipFree(c->dGTabs); c->dGTabs = nullptr; c->gtabsCap = 0;
                if (hipMalloc((void**)&c->dGTabs, gridG * gtabWords * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); gridG = 0; }
                else c->gtabsCap = gridG * gtabWords;
            }
        }
        if (gridG) HIPCHK(c, hipEventRecord(c->coEv[0], s));
        hipLaunchKernelGGL(zhip::k_parse_fast_q, dim3((unsigned)gridQ), dim3(64), smem, s,
                           srcDev, c->dUnits, c->dSlots, (uint32_t)nUnits, c->dSeqs, c->dLits, c->dParse, order, c->dQueue);
        if (gridG) {
            HIPCHK(c, hipStreamWaitEvent(c->coStream, c->coEv[0], 0));
            hipLaunchKernelGGL(zhip::k_parse_fast_g, dim3((unsigned)gridG), dim3(64), 0, c->coStream,
                               srcDev, c->dUnits, c->dSlots, (uint32_t)nUnits, c->dSeqs, c->dLits, c->dParse, order, c->dQueue, c->dGTabs, gtabWords);
            HIPCHK(c, hipEventRecord(c->coEv[1], c->coStream));
            HIPCHK(c, hipStreamWaitEvent(s, c->coEv[1], 0));
        }
    } else if (c->strategy & 1)
        hipLaunchKernelGGL(zhip::k_parse_fast, dim3((unsigned)nUnits), dim3(64), smem, s,
                           srcDev, c->dUnits, c->dSlots, (uint32_t)nUnits, c->dSeqs, c->dLits, c->dParse);
    if (c->strategy & 4) {
        c->hcEvUsed = 0;
        for (size_t u0 = 0; u0 < nUnits; u0 += c->hcChunk) {
            unsigned const nu = (unsigned)(nUnits - u0 < c->hcChunk ? nUnits - u0 : c->hcChunk);
            while (c->hcEv.size() < c->hcEvUsed + 4) { hipEvent_t e; HIPCHK(c, hipEventCreate(&e)); c->hcEv.push_back(e); }
            hipEvent_t* const he = &c->hcEv[c->hcEvUsed]; c->hcEvUsed += 4;
            HIPCHK(c, hipEventRecord(he[0], s));
            {   // the builder's LDS covers the row counters, tag filters and the 2 304-byte stage as well (rh_chain_lds_need): 69 952 bytes at an effective hashLog of 18
                // (a cparams override on units above 64 KB) — past the 64 KB a launch may ask for without the attribute
                size_t const ldsC = zhip::hc_chain_lds_bytes(c->hcHashLog);
                if (ldsC > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)zhip::k_hc_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsC));
                hipLaunchKernelGGL(zhip::k_hc_chain, dim3(nu), dim3(64), ldsC, s,
                                   srcDev, c->dUnits + u0, nu, c->dTabs, c->tabStride, c->dBest);
                HIPCHK(c, hipGetLastError());
            }
            HIPCHK(c, hipEventRecord(he[1], s));
            {   size_t const lds = (((size_t)c->hcMaxLen + 15) & ~(size_t)15) + 32;
                if (lds > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)zhip::k_hc_search_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(zhip::k_hc_search_lds, dim3(nu), dim3(ZHIP_HC_SEARCH_LDS_THREADS), lds, s,
                                   srcDev, c->dUnits + u0, nu, c->dTabs, c->tabStride, c->dBest, (const ZhipParse*)nullptr);
            }
            HIPCHK(c, hipEventRecord(he[2], s));
            {   // row matcher: the two-pass prediction of the positions the 384-position rule leaves out (zhip_parse_lazy.h: rh_reconcile).  The
                // parse is TRIED first; a unit whose parse had to redo more than the budget of searches live gives up, and only those
                // units get the predicting parse, their records again without the predicted positions, and the parse again.  Data without
                // long matches never leaves the first launch (zhip_set_prediction(units = 1) turns it on; the budget is 256 live searches)
                int const predictOn = c->rhPredict;          // default off for units (zhip_set_prediction): exact either way (emulator, GPU parity tests), DESIGN.md 4.2b
                int const budget = 256;
                bool anyRow = false;
                for (uint32_t i = 0; i < nu && !anyRow; i++) anyRow = c->hUnits[u0 + i].rowLog != 0;
                if (predictOn && anyRow && budget > 0) {
                    size_t const lds = (((size_t)c->hcMaxLen + 15) & ~(size_t)15) + 32;
                    hipLaunchKernelGGL(zhip::k_parse_lazy, dim3(nu), dim3(64), ZHIP_RH_DIRTY_BYTES, s,
                                       srcDev, c->dUnits + u0, c->dSlots + u0, nu, c->dTabs, c->tabStride, c->dBest, c->dSeqs, c->dLits, c->dParse + u0, 2u, (uint32_t)budget);
                    hipLaunchKernelGGL(zhip::k_parse_lazy, dim3(nu), dim3(64), ZHIP_RH_DIRTY_BYTES, s,
                                       srcDev, c->dUnits + u0, c->dSlots + u0, nu, c->dTabs, c->tabStride, c->dBest, c->dSeqs, c->dLits, c->dParse + u0, 1u, 0u);
                    hipLaunchKernelGGL(zhip::k_hc_search_lds, dim3(nu), dim3(ZHIP_HC_SEARCH_LDS_THREADS), lds, s,
                                       srcDev, c->dUnits + u0, nu, c->dTabs, c->tabStride, c->dBest, (const ZhipParse*)(c->dParse + u0));
                    hipLaunchKernelGGL(zhip::k_parse_lazy, dim3(nu), dim3(64), ZHIP_RH_DIRTY_BYTES, s,
                                       srcDev, c->dUnits + u0, c->dSlots + u0, nu, c->dTabs, c->tabStride, c->dBest, c->dSeqs, c->dLits, c->dParse + u0, 3u, 0u);
                } else
                hipLaunchKernelGGL(zhip::k_parse_lazy, dim3(nu), dim3(64), ZHIP_RH_DIRTY_BYTES, s,
                                   srcDev, c->dUnits + u0, c->dSlots + u0, nu, c->dTabs, c->tabStride, c->dBest,
                                   c->dSeqs, c->dLits, c->dParse + u0, 0u, 0u);
            }
            HIPCHK(c, hipEventRecord(he[3], s));
        }
    } else c->hcEvUsed = 0;
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    return 0;
}

// stages 2 + 3 on stream s; dstDev receives the packed frames
static size_t launch_entropy_gather(zhip_ctx* c, const uint8_t* srcDev, size_t nUnits, uint8_t* dstDev, hipStream_t s)
{
    static bool attrSet = false;
    if (!attrSet) { (void)hipFuncSetAttribute((const void*)zhip::k_entropy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(zhip::EntShared)); attrSet = true; }
    if (c->checksum) {                          // ZSTD_c_checksumFlag: XXH64 of every unit's content, 16 units per wavefront
        if (!c->dChecks) HIPCHK(c, hipMalloc((void**)&c->dChecks, (c->maxUnits + 16) * sizeof(uint32_t)));
        hipLaunchKernelGGL(zhip::k_xxh64, dim3((unsigned)((nUnits + 15) / 16)), dim3(64), 0, s, srcDev, c->dUnits, (uint32_t)nUnits, c->dChecks);
    }
    {   // units of at most ZHIP_ENT_SMALL_MAX bytes take the one-wavefront form of the encoder (small records: a 256-thread
        // workgroup would mostly wait at its own barriers)
        int const useSmall = 1;
        bool anySmall = false, anyLarge = false;
        if (useSmall) for (size_t i = 0; i < nUnits && !(anySmall && anyLarge); i++) { if (c->hUnits[i].srcLen <= ZHIP_ENT_SMALL_MAX) anySmall = true; else anyLarge = true; }
        else anyLarge = true;
        const uint32_t* const ck = c->checksum ? c->dChecks : (const uint32_t*)nullptr;
        if (anyLarge)
            hipLaunchKernelGGL(zhip::k_entropy, dim3((unsigned)nUnits), dim3(ZHIP_ENT_THREADS), sizeof(zhip::EntShared), s,
                               srcDev, c->dUnits, c->dSlots, (uint32_t)nUnits, c->dSeqs, c->dParse, c->dLits, c->dStBits, c->dOut, c->dOutSize, c->curDictEntropy, c->curDictID, ck, anySmall ? 1u : 0u);
        if (anySmall)
            hipLaunchKernelGGL(zhip::k_entropy_small, dim3((unsigned)nUnits), dim3(64), sizeof(zhip::EntSharedSmall) + ZHIP_ENT_SMALL_PAD, s,
                               srcDev, c->dUnits, c->dSlots, (uint32_t)nUnits, c->dSeqs, c->dParse, c->dLits, c->dStBits, c->dOut, c->dOutSize, c->curDictEntropy, c->curDictID, ck);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[2], s));
    {   size_t const e_ = launch_offsets(c, nUnits, s); if (zhip_isError(e_)) return e_; }
    hipLaunchKernelGGL(zhip::k_gather, dim3((unsigned)nUnits), dim3(256), 0, s, c->dOut, c->dSlots, c->dOutSize, c->dOutOff, (uint32_t)nUnits, dstDev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[3], s));
    return 0;
}

// chunked variant of launch_parse + launch_entropy_gather (see zhip_ctx_s::nChunks).  ev[1] is not meaningful here:
// the stages of different chunks overlap, so timing[] reports the match finder + entropy time as one figure.
static size_t launch_pipelined(zhip_ctx* c, const uint8_t* srcDev, size_t nUnits, uint32_t maxHashLog, uint8_t* dstDev, hipStream_t s)
{
    size_t const smem = zhip::fast_tag_lds_bytes(maxHashLog);
    static bool attrSet = false;
    if (!attrSet) { (void)hipFuncSetAttribute((const void*)zhip::k_entropy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(zhip::EntShared)); attrSet = true; }
    if (smem > 64 * 1024)
        HIPCHK(c, hipFuncSetAttribute((const void*)zhip::k_parse_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    HIPCHK(c, hipMemcpyAsync(c->dUnits, c->hUnits, nUnits * sizeof(ZhipUnit), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->dSlots, c->hSlots, nUnits * sizeof(ZhipSlot), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    size_t const per = (nUnits + c->nChunks - 1) / c->nChunks;
    for (int i = 0; i < c->nChunks; i++) {
        size_t const u0 = (size_t)i * per, u1 = u0 + per < nUnits ? u0 + per : nUnits;
        if (u0 >= u1) break;
        unsigned const nu = (unsigned)(u1 - u0);
        hipStream_t q = c->cs[i];
        HIPCHK(c, hipStreamWaitEvent(q, c->ev[0], 0));
        hipLaunchKernelGGL(zhip::k_parse_fast, dim3(nu), dim3(64), smem, q, srcDev, c->dUnits + u0, c->dSlots + u0, nu,
                           c->dSeqs, c->dLits, c->dParse + u0);
        hipLaunchKernelGGL(zhip::k_entropy, dim3(nu), dim3(ZHIP_ENT_THREADS), sizeof(zhip::EntShared), q,
                           srcDev, c->dUnits + u0, c->dSlots + u0, nu, c->dSeqs, c->dParse + u0, c->dLits,
                           c->dStBits, c->dOut, c->dOutSize + u0, (const zhip::ZhipDictEntropy*)nullptr, 0u, (const uint32_t*)nullptr, 0u);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(c->cev[i], q));
        HIPCHK(c, hipStreamWaitEvent(s, c->cev[i], 0));
    }
    HIPCHK(c, hipEventRecord(c->ev[2], s));
    {   size_t const e_ = launch_offsets(c, nUnits, s); if (zhip_isError(e_)) return e_; }
    hipLaunchKernelGGL(zhip::k_gather, dim3((unsigned)nUnits), dim3(256), 0, s, c->dOut, c->dSlots, c->dOutSize, c->dOutOff, (uint32_t)nUnits, dstDev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[3], s));
    return 0;
}

static void read_timing(zhip_ctx* c)
{
    float a = 0, b = 0, g = 0, tot = 0;
    (void)hipEventElapsedTime(&a, c->ev[0], c->ev[1]);
    (void)hipEventElapsedTime(&b, c->ev[1], c->ev[2]);
    (void)hipEventElapsedTime(&g, c->ev[2], c->ev[3]);
    (void)hipEventElapsedTime(&tot, c->ev[0], c->ev[3]);
    c->timing[0] = a; c->timing[1] = b; c->timing[2] = g; c->timing[3] = tot;
}

// device pipeline; returns total compressed size (after synchronising s)
static size_t units_as_frames_locked(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, const unsigned long long* offs,
                                     size_t nFrames, int level, uint32_t* frameSizesDev, hipStream_t s);        // = frames_device_locked, below
static size_t compress_device_locked(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, size_t srcSize,
                                     int level, size_t unitSize, uint32_t* unitSizesDev, hipStream_t s)
{
    size_t err = 0; uint32_t mh = 0;
    c->wideFast = false;
    size_t const nUnits = build_units(c, srcSize, unitSize, level, &err, &mh);
    if (!nUnits && c->wideFast) {
        // explicit parameters with ZSTD_fast and hashLog > 15: every unit is a one-block frame for the frame kernel (k_frame_fast / k_frame_hbm:
        // 24-bit LDS or 32-bit HBM table by size, zhip_frame.h) — the same bytes, the reference's ZSTD_compress2 of the unit
        c->wideFast = false;
        size_t const nU = srcSize ? (srcSize + unitSize - 1) / unitSize : 1;
        if (nU > c->maxUnits) return ZERR(ZE_srcSize_wrong);
        if (dstCapacity < zhip_compressBound(srcSize, unitSize)) return ZERR(ZE_dstSize_tooSmall);
        std::vector<unsigned long long> offs(nU + 1);
        for (size_t i = 0; i <= nU; i++) offs[i] = i * unitSize < srcSize ? (unsigned long long)(i * unitSize) : (unsigned long long)srcSize;
        return units_as_frames_locked(c, dstDev, dstCapacity, srcDev, offs.data(), nU, level, unitSizesDev, s);
    }
    if (!nUnits) return err;
    if (dstCapacity < zhip_compressBound(srcSize, unitSize)) return ZERR(ZE_dstSize_tooSmall);
    size_t r;
    if (c->nChunks > 1 && c->strategy == 1 && !c->checksum && nUnits >= (size_t)64 * c->nChunks) r = launch_pipelined(c, (const uint8_t*)srcDev, nUnits, mh, (uint8_t*)dstDev, s);
    else {
        r = launch_parse(c, (const uint8_t*)srcDev, nUnits, mh, s);
        if (zhip_isError(r)) return r;
        r = launch_entropy_gather(c, (const uint8_t*)srcDev, nUnits, (uint8_t*)dstDev, s);
    }
    if (zhip_isError(r)) return r;
    if (unitSizesDev) HIPCHK(c, hipMemcpyAsync(unitSizesDev, c->dOutSize, nUnits * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    uint64_t total = 0;
    HIPCHK(c, hipMemcpyAsync(c->hOutSize, c->dOutSize, nUnits * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    for (size_t i = 0; i < nUnits; i++) total += c->hOutSize[i];
    read_timing(c);
    c->stats[0] = nUnits; c->stats[1] = srcSize; c->stats[2] = total; c->stats[3] = 0; c->stats[4] = 0;
    c->nUnits = nUnits;
    return (size_t)total;
}

extern "C" {

size_t zhip_compress_device(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, size_t srcSize,
                            int level, size_t unitSize, uint32_t* unitSizesDev, void* stream)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return compress_device_locked(c, dstDev, dstCapacity, srcDev, srcSize, level, unitSize, unitSizesDev,
                                  stream ? (hipStream_t)stream : c->stream);
}

// explicit compression parameters (ZSTD_c_windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy; 0 = the
// level's own): set for the duration of one call
struct OvrScope {
    zhip_ctx* c;
    OvrScope(zhip_ctx* c_, const unsigned* cp) : c(c_) { if (cp) { memcpy(c->ovr, cp, sizeof(c->ovr)); c->haveOvr = true; } }
    ~OvrScope() { c->haveOvr = false; }
};
size_t zhip_compress_params_device(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, size_t srcSize,
                                   int level, const unsigned cparams[7], size_t unitSize, uint32_t* unitSizesDev, void* stream)
{
    if (cparams && !zhip::host_check_overrides(cparams)) return ZERR(ZE_parameter_outOfBound);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    OvrScope scope(c, cparams);
    return compress_device_locked(c, dstDev, dstCapacity, srcDev, srcSize, level, unitSize, unitSizesDev, stream ? (hipStream_t)stream : c->stream);
}
static size_t compress_host_locked(zhip_ctx* c, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, size_t unitSize, size_t* unitSizes);
size_t zhip_compress_params(zhip_ctx* c, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                            int level, const unsigned cparams[7], size_t unitSize, size_t* unitSizes)
{
    if (cparams && !zhip::host_check_overrides(cparams)) return ZERR(ZE_parameter_outOfBound);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    OvrScope scope(c, cparams);
    return compress_host_locked(c, dst, dstCapacity, src, srcSize, level, unitSize, unitSizes);
}
int zhip_getCParams_explicit(int level, unsigned long long srcSize, const unsigned cparams[7], unsigned out[7])
{
    zhip::CParams cp;
    if (cparams && !zhip::host_check_overrides(cparams)) return 2;
    if (!zhip::host_get_cparams(level, srcSize, &cp, cparams)) return 1;
    out[0] = cp.windowLog; out[1] = cp.chainLog; out[2] = cp.hashLog; out[3] = cp.searchLog; out[4] = cp.minMatch; out[5] = cp.targetLength; out[6] = cp.strategy;
    return 0;
}

size_t zhip_compress(zhip_ctx* c, void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                     int level, size_t unitSize, size_t* unitSizes)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return compress_host_locked(c, dst, dstCapacity, src, srcSize, level, unitSize, unitSizes);
}
static size_t compress_host_locked(zhip_ctx* c, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, size_t unitSize, size_t* unitSizes)
{
    size_t const bound = zhip_compressBound(srcSize, unitSize);
    if (dstCapacity < bound) return ZERR(ZE_dstSize_tooSmall);
    if (c->srcStageCap < srcSize + 64) {
        (void)hipFree(c->dSrcStage); c->dSrcStage = nullptr; c->srcStageCap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dSrcStage, srcSize + 64)); c->srcStageCap = srcSize + 64;
    }
    if (c->dstStageCap < bound + 64) {
        (void)hipFree(c->dDstStage); c->dDstStage = nullptr; c->dstStageCap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dDstStage, bound + 64)); c->dstStageCap = bound + 64;
    }
    if (srcSize) HIPCHK(c, hipMemcpyAsync(c->dSrcStage, src, srcSize, hipMemcpyHostToDevice, c->stream));
    size_t const total = compress_device_locked(c, c->dDstStage, c->dstStageCap, c->dSrcStage, srcSize, level, unitSize, nullptr, c->stream);
    if (zhip_isError(total)) return total;
    HIPCHK(c, hipMemcpy(dst, c->dDstStage, total, hipMemcpyDeviceToHost));
    if (unitSizes) for (size_t i = 0; i < c->nUnits; i++) unitSizes[i] = c->hOutSize[i];
    return total;
}

// ------------------------------------------------------------------ multi-block frames
// Each input becomes ONE standard frame holding the same blocks ZSTD_compress2 emits for it (zstd_compress.c:4520-4640): the
// match finder's table, the window, the repcodes and the literals' Huffman table carry over from block to block.  The chain
// is serial inside a frame (one workgroup per frame, zhip_frame.h); the frames of a batch run concurrently.
// With mt (ZSTD_c_nbWorkers >= 1 semantics, zstdmt_compress.c): an input above 512 KB is cut into jobs of the job size; each job is
// compressed by its own workgroup with the overlap in front of it as prefix, and the concatenation is the frame the reference's
// worker pool emits (it does not depend on the number of workers).  A unit of the launch is then a job, not a frame.
struct MtParams { bool on; unsigned long long jobSize; int overlapLog; };
// ZSTD_fast frames whose table fits LDS run two workgroups per CU (k_frame_fast); a batch of more workgroups than that holds at once goes to the all-HBM form instead — four per
// CU, each parser somewhat slower on its table in L2 / HBM, half as many rounds (round 6, profiles/r06_ab_frames_hbm_tables.log: 1 024 frames of 1 MiB, datagen 45.5 -> 39.9 ms,
// text 191.7 -> 121.4 ms; 256 frames: 20.5 vs 26.2 ms the other way).  tabWordsAll = the largest table of the batch's ZSTD_fast / ZSTD_dfast workgroups.
static void frames_prefer_hbm(const zhip_ctx* c, size_t nU, uint32_t& ldsTab, size_t& tabWords, size_t tabWordsAll)
{
    if (!ldsTab || nU <= (size_t)2 * (size_t)c->numCUs) return;
    if (nU * tabWordsAll * sizeof(uint32_t) > ((size_t)2 << 30)) return;          // (tables are per workgroup, not per resident workgroup, in the frame kernels)
    ldsTab = 0;
    if (tabWordsAll > tabWords) tabWords = tabWordsAll;
}
static size_t frames_run_locked(zhip_ctx* c, void* dstDev, const void* srcDev, size_t nU, size_t nFrames, uint32_t ldsTab, size_t tabWords,
                                size_t outBytes, unsigned long long totalSrc, bool withJobs, uint32_t* frameSizesDev, hipStream_t s);
static size_t frames_device_locked(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, const unsigned long long* offs,
                                   size_t nFrames, int level, uint32_t* frameSizesDev, hipStream_t s, MtParams mt = MtParams{false, 0, 0})
{
    if (nFrames == 0) return ZERR(ZE_srcSize_wrong);
    if (nFrames > c->maxUnits) { snprintf(c->err, sizeof(c->err), "%zu frames > context capacity %zu", nFrames, c->maxUnits); return ZERR(ZE_srcSize_wrong); }
    const unsigned* const ov = c->haveOvr ? c->ovr : nullptr;
    size_t bound = 0, outBytes = 0, tabWords = 0, tabWordsAll = 0, nU = 0; uint32_t ldsTab = 0; unsigned long long totalSrc = 0;
    if (mt.on) { c->hJobs.clear(); c->hFrameUnits.resize(nFrames); }
    bool lzAny = false, lzAll = true;
    c->hLz.clear(); c->lzPos = 0; c->lzHeads = 0; c->lzRing = 0; c->lzLongest = 1;
    for (size_t i = 0; i < nFrames; i++) {
        if (offs[i + 1] < offs[i] || offs[i + 1] - offs[i] >= (1ull << 31)) { snprintf(c->err, sizeof(c->err), "frame %zu: inputs of 2 GiB and more are not implemented on device", i); return ZERR(ZE_srcSize_wrong); }
        size_t const n = (size_t)(offs[i + 1] - offs[i]);
        zhip::CParams cp;
        if (!zhip::host_get_cparams(level, n, &cp, ov)) return ZERR(ZE_parameter_unsupported);
        if (cp.strategy < ZHIP_STRAT_FAST || cp.strategy > ZHIP_STRAT_LAZY2) { snprintf(c->err, sizeof(c->err), "multi-block frames: strategy %u not implemented on device (ZSTD_fast ... ZSTD_lazy2 only)", cp.strategy); return ZERR(ZE_parameter_unsupported); }
        bool const lazy = cp.strategy >= ZHIP_STRAT_GREEDY;
        uint32_t rowLog = 0;                                                                          // zstd_compress.c:237-253, :2042
        if (lazy && c->rowMode != 2 && cp.windowLog > 14) rowLog = cp.searchLog < 4 ? 4 : (cp.searchLog > 6 ? 6 : cp.searchLog);
        else if (lazy && c->rowMode == 1) { snprintf(c->err, sizeof(c->err), "ZSTD_ps_enable with windowLog %u <= 14: no row matcher for it on device", cp.windowLog); return ZERR(ZE_parameter_unsupported); }
        if (lazy) lzAny = true; else lzAll = false;
        if (cp.windowLog < 17 && n > ((size_t)1 << cp.windowLog)) { snprintf(c->err, sizeof(c->err), "multi-block frames: windowLog %u below the block size is not implemented on device", cp.windowLog); return ZERR(ZE_parameter_unsupported); }
        // the sections of this frame: one without workers or at most 512 KB (ZSTDMT_JOBSIZE_MIN: the reference then runs single-threaded)
        size_t section = n ? n : 1, overlap = 0;
        if (mt.on && n > zhip::MT_JOBSIZE_MIN) {
            section = zhip::host_mt_job_size(cp, mt.jobSize); overlap = zhip::host_mt_overlap_size(cp, mt.overlapLog);
            if (section < overlap) section = overlap;                                                   // zstdmt_compress.c:1300
        }
        size_t const nSec = n ? (n + section - 1) / section : 1;
        if (nU + nSec > c->maxUnits) { snprintf(c->err, sizeof(c->err), "%zu jobs > context capacity %zu", nU + nSec, c->maxUnits); return ZERR(ZE_srcSize_wrong); }
        if ((nU + nSec) * (size_t)ZHIP_SEQ_CAP > c->seqArena || (nU + nSec) * (size_t)ZHIP_LIT_STRIDE > c->litArena) {      // a context made for small records has smaller arenas
            snprintf(c->err, sizeof(c->err), "%zu jobs need full-size slots: this context's arenas are smaller (created for records?)", nU + nSec); return ZERR(ZE_srcSize_wrong); }
        size_t prevLen = 0;
        for (size_t k = 0; k < nSec; k++, nU++) {
            size_t const start = k * section, len = n - start < section ? n - start : section;
            ZhipUnit& u = c->hUnits[nU];
            u.srcOff = offs[i]; u.srcLen = (uint32_t)len;
            u.windowLog = (uint8_t)cp.windowLog; u.chainLog = (uint8_t)cp.chainLog; u.hashLog = (uint8_t)cp.hashLog;
            u.minMatch = (uint8_t)cp.minMatch; u.strategy = (uint8_t)cp.strategy; u.searchLog = (uint8_t)cp.searchLog;
            u.litMode = (cp.strategy == ZHIP_STRAT_FAST && cp.targetLength > 0) ? 1 : 0; u.pad0 = 0; u.targetLength = cp.targetLength; u.rowLog = rowLog; u.pad1 = 0;
            ZhipSlot& sl = c->hSlots[nU];
            sl.seqOff = nU * (uint64_t)ZHIP_SEQ_CAP; sl.litOff = nU * (uint64_t)ZHIP_LIT_STRIDE; sl.outOff = outBytes; sl.seqCap = ZHIP_SEQ_CAP; sl.pad0 = 0;
            outBytes += (zhip::host_compress_bound(len) + 1024 + 15) & ~(size_t)15;      // the block in flight may overshoot before it is declared raw
            {   zhip::ZhipLzSlot L; memset(&L, 0, sizeof(L));
                size_t const pre = (mt.on && k) ? (prevLen < overlap ? prevLen : overlap) : 0;
                if (lazy && pre + len >= ((size_t)1 << 30)) {             // a link is 30 bits + two flags (zhip_frame_lazy.h: ZHIP_LZ_LINK)
                    snprintf(c->err, sizeof(c->err), "frame %zu: a window of 1 GiB and more with a lazy strategy is not implemented on device (use ZSTD_c_nbWorkers: jobs)", i); return ZERR(ZE_srcSize_wrong); }
                if (lazy) { zhip::lz_fill_slot(L, u, (uint32_t)pre, c->lzPos, c->lzHeads, c->lzRing); if (len > c->lzLongest) c->lzLongest = (uint32_t)len; }
                c->hLz.push_back(L);
            }
            if (mt.on) {
                zhip::ZhipJob j;
                j.start = (uint32_t)start; j.prefixLen = (uint32_t)(k == 0 ? 0 : (prevLen < overlap ? prevLen : overlap));    // zstdmt_compress.c:1404-1407
                j.flags = (k == 0 ? ZHIP_JOB_FIRST : 0u) | (k + 1 == nSec ? ZHIP_JOB_LAST : 0u);
                j.ownHeader = zhip::frame_header_bytes_multi((uint32_t)len, cp.windowLog); j.frameSize = n; j.frameIdx = (uint32_t)i; j.pad0 = 0;
                c->hJobs.push_back(j);
            }
            prevLen = len;
        }
        if (mt.on) { ZhipUnit& fu = c->hFrameUnits[i]; fu = c->hUnits[nU - 1]; fu.srcLen = (uint32_t)n; }
        bound += zhip::host_compress_bound(n);
        if (!lazy) {   // the longest walk of this frame: a section plus its prefix (the whole input without jobs)
            uint32_t const mode = zhip::frame_table_mode(cp.strategy, cp.hashLog, (unsigned long long)(n < section ? n : section) + overlap + 1);
            size_t const w = zhip::frame_table_words(cp.strategy, cp.hashLog, cp.chainLog);
            if (w > tabWordsAll) tabWordsAll = w;
            if (mode == zhip::ZHIP_FT_HBM) { if (w > tabWords) tabWords = w; }
            else { uint32_t const b = zhip::frame_table_lds_bytes(mode, cp.hashLog); if (b > ldsTab) ldsTab = b; }
        }
        totalSrc += n;
    }
    if (dstCapacity < bound) return ZERR(ZE_dstSize_tooSmall);
    c->lzAny = lzAny; c->lzAll = lzAny && lzAll;
    frames_prefer_hbm(c, nU, ldsTab, tabWords, tabWordsAll);
    size_t const r = frames_run_locked(c, dstDev, srcDev, nU, nFrames, ldsTab, tabWords, outBytes, totalSrc, mt.on, frameSizesDev, s);
    c->lzAny = c->lzAll = false;
    return r;
}

// the launch part: c->hUnits / c->hSlots (and c->hJobs / c->hFrameUnits with jobs) describe nU workgroups of nFrames frames
static size_t frames_run_locked(zhip_ctx* c, void* dstDev, const void* srcDev, size_t nU, size_t nFrames, uint32_t ldsTab, size_t tabWords,
                                size_t outBytes, unsigned long long totalSrc, bool withJobs, uint32_t* frameSizesDev, hipStream_t s)
{
    struct { bool on; } mt = { withJobs };
    if (c->frameOutCap < outBytes) {
        (void)hipFree(c->dFrameOut); c->dFrameOut = nullptr; c->frameOutCap = 0;
        if (hipMalloc((void**)&c->dFrameOut, outBytes) != hipSuccess) { snprintf(c->err, sizeof(c->err), "cannot allocate %zu bytes of frame output room", outBytes); return ZERR(ZE_memory_allocation); }
        c->frameOutCap = outBytes;
    }
    if (c->frameStateCap < nU) {
        (void)hipFree(c->dFrameState); c->dFrameState = nullptr; c->frameStateCap = 0;
        if (hipMalloc((void**)&c->dFrameState, nU * sizeof(zhip::ZhipFrameState)) != hipSuccess) return ZERR(ZE_memory_allocation);
        c->frameStateCap = nU;
    }
    size_t const tabStride = (tabWords + 3) & ~(size_t)3;
    if (tabStride && c->tabsCap < nU * tabStride) {
        (void)hipFree(c->dTabs); c->dTabs = nullptr; c->tabsCap = 0;
        if (hipMalloc((void**)&c->dTabs, nU * tabStride * sizeof(uint32_t)) != hipSuccess) return ZERR(ZE_memory_allocation);
        c->tabsCap = nU * tabStride;
    }
    if (mt.on) {
        if (c->jobsCap < nU) {
            (void)hipFree(c->dJobs); c->dJobs = nullptr; c->jobsCap = 0;
            if (hipMalloc((void**)&c->dJobs, nU * sizeof(zhip::ZhipJob)) != hipSuccess) return ZERR(ZE_memory_allocation);
            c->jobsCap = nU;
        }
        if (c->frameUnitsCap < nFrames) {
            (void)hipFree(c->dFrameUnits); (void)hipFree(c->dFrameSizes); c->dFrameUnits = nullptr; c->dFrameSizes = nullptr; c->frameUnitsCap = 0;
            if (hipMalloc((void**)&c->dFrameUnits, nFrames * sizeof(ZhipUnit)) != hipSuccess || hipMalloc((void**)&c->dFrameSizes, nFrames * sizeof(uint32_t)) != hipSuccess) return ZERR(ZE_memory_allocation);
            c->frameUnitsCap = nFrames;
        }
        HIPCHK(c, hipMemcpyAsync(c->dJobs, c->hJobs.data(), nU * sizeof(zhip::ZhipJob), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->dFrameUnits, c->hFrameUnits.data(), nFrames * sizeof(ZhipUnit), hipMemcpyHostToDevice, s));
    }
    if (c->lzAny) {
        // per position of every window: link (4 B) + tag (1 B) + record (16 B); per unit a head table of 4 << keyBits bytes
        size_t const needPos = (size_t)c->lzPos + 64, needHeads = (size_t)c->lzHeads + 64;
        if (c->lzPosCap < needPos) {
            (void)hipFree(c->dLzPrev); (void)hipFree(c->dLzTags); (void)hipFree(c->dLzBest); c->dLzPrev = nullptr; c->dLzTags = nullptr; c->dLzBest = nullptr; c->lzPosCap = 0;
            if (hipMalloc((void**)&c->dLzPrev, needPos * sizeof(uint32_t)) != hipSuccess || hipMalloc((void**)&c->dLzTags, needPos) != hipSuccess ||
                hipMalloc((void**)&c->dLzBest, needPos * sizeof(zhip::LzRec)) != hipSuccess) {
                (void)hipGetLastError();
                snprintf(c->err, sizeof(c->err), "lazy frames: cannot allocate %zu bytes of match-state room (21 bytes per position)", needPos * 21); return ZERR(ZE_memory_allocation); }
            c->lzPosCap = needPos;
        }
        if (c->lzHeadCap < needHeads) {
            (void)hipFree(c->dLzHeads); c->dLzHeads = nullptr; c->lzHeadCap = 0;
            if (hipMalloc((void**)&c->dLzHeads, needHeads * sizeof(uint32_t)) != hipSuccess) { (void)hipGetLastError(); return ZERR(ZE_memory_allocation); }
            c->lzHeadCap = needHeads;
        }
        if (c->lzSlotCap < nU) {
            (void)hipFree(c->dLzSlots); c->dLzSlots = nullptr; c->lzSlotCap = 0;
            if (hipMalloc((void**)&c->dLzSlots, nU * sizeof(zhip::ZhipLzSlot)) != hipSuccess) {
                (void)hipGetLastError();
                (void)hipFree(c->dLzRing); c->dLzRing = nullptr; c->lzRingCap = 0;                 // the optional arena makes room for the mandatory one
                if (hipMalloc((void**)&c->dLzSlots, nU * sizeof(zhip::ZhipLzSlot)) != hipSuccess) { (void)hipGetLastError(); return ZERR(ZE_memory_allocation); }
            }
            c->lzSlotCap = nU;
        }
        // the OPTIONAL arena last: the live rows must never take the room a mandatory buffer of this call needs
        if (c->lzRingOn && c->lzRing && c->lzRingCap < (size_t)c->lzRing + 256) {
            (void)hipFree(c->dLzRing); c->dLzRing = nullptr; c->lzRingCap = 0;
            if (hipMalloc((void**)&c->dLzRing, (size_t)c->lzRing + 256) == hipSuccess) c->lzRingCap = (size_t)c->lzRing + 256;
            else (void)hipGetLastError();                         // no room for the live rows: the parser walks the links instead (same bytes, slower on long matches)
        }
        HIPCHK(c, hipMemcpyAsync(c->dLzSlots, c->hLz.data(), nU * sizeof(zhip::ZhipLzSlot), hipMemcpyHostToDevice, s));
    }
    size_t const lds = zhip::frame_lds_bytes(ldsTab);
    bool const hbmOnly = ldsTab == 0;                                                  // no table in LDS: the variant compiled for four workgroups per CU
    HIPCHK(c, hipFuncSetAttribute(hbmOnly ? (const void*)zhip::k_frame_hbm : (const void*)zhip::k_frame_fast, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(c, hipMemcpyAsync(c->dUnits, c->hUnits, nU * sizeof(ZhipUnit), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->dSlots, c->hSlots, nU * sizeof(ZhipSlot), hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(c->ev[0], s));
    if (c->checksum) {                                                                    // of whole frames, whatever the jobs
        if (!c->dChecks) HIPCHK(c, hipMalloc((void**)&c->dChecks, (c->maxUnits + 16) * sizeof(uint32_t)));
        hipLaunchKernelGGL(zhip::k_xxh64_wave, dim3((unsigned)nFrames), dim3(64), ZHIP_XXH_WAVE_LDS, s, (const uint8_t*)srcDev, mt.on ? c->dFrameUnits : c->dUnits, (uint32_t)nFrames, c->dChecks);
    }
    HIPCHK(c, hipEventRecord(c->ev[1], s));
    if (c->lzAny) {
        const zhip::ZhipJob* const jb = mt.on ? c->dJobs : (const zhip::ZhipJob*)nullptr;
        HIPCHK(c, hipFuncSetAttribute((const void*)zhip::k_frame_lazy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)zhip::frame_lazy_lds_bytes()));
        hipLaunchKernelGGL(zhip::k_lz_links, dim3((unsigned)nU), dim3(ZHIP_LZ_LINK_THREADS), sizeof(zhip::LzLinkShared), s,
                           (const uint8_t*)srcDev, c->dUnits, jb, c->dLzSlots, (uint32_t)nU, c->dLzPrev, c->dLzTags, c->dLzHeads);
        for (size_t w0 = 0; w0 < nU; w0 += 32768) {
            size_t const nw = nU - w0 < 32768 ? nU - w0 : 32768;
            hipLaunchKernelGGL(zhip::k_lz_search, dim3((c->lzLongest + 255) / 256, (unsigned)nw), dim3(256), 0, s,
                               (const uint8_t*)srcDev, c->dUnits, jb, c->dLzSlots, (uint32_t)w0, (uint32_t)nU, c->dLzPrev, c->dLzTags, c->dLzBest, (const zhip::ZhipFrameState*)nullptr);
        }
        {   // the two-pass prediction of the positions the parse leaves un-inserted (zhip_frame_lazy.h: frame_lazy_predict)
            int const predictOn = c->lzPredict;              // on by default (zhip_set_prediction): a window whose first 32 KB leave nothing out is parsed once
            if (predictOn) {
                hipLaunchKernelGGL(zhip::k_lz_predict, dim3((unsigned)nU), dim3(64), sizeof(ZhipParse), s,
                                   (const uint8_t*)srcDev, c->dUnits, jb, c->dLzSlots, (uint32_t)nU, c->dLzPrev, c->dLzTags, c->dLzBest, c->dFrameState);
                for (size_t w0 = 0; w0 < nU; w0 += 32768) {
                    size_t const nw = nU - w0 < 32768 ? nU - w0 : 32768;
                    hipLaunchKernelGGL(zhip::k_lz_search, dim3((c->lzLongest + 255) / 256, (unsigned)nw), dim3(256), 0, s,
                                       (const uint8_t*)srcDev, c->dUnits, jb, c->dLzSlots, (uint32_t)w0, (uint32_t)nU, c->dLzPrev, c->dLzTags, c->dLzBest, (const zhip::ZhipFrameState*)c->dFrameState);
                }
            }
        }
        hipLaunchKernelGGL(zhip::k_frame_lazy, dim3((unsigned)nU), dim3(ZHIP_ENT_THREADS), zhip::frame_lazy_lds_bytes(), s,
                           (const uint8_t*)srcDev, c->dUnits, c->dSlots, jb, c->dLzSlots, (uint32_t)nU, c->dLzPrev, c->dLzTags, c->dLzBest, c->dLzHeads,
                           (c->lzRingOn && c->lzRingCap >= (size_t)c->lzRing + 256) ? c->dLzRing : (uint8_t*)nullptr, c->dSeqs, c->dLits, c->dStBits, c->dFrameOut, c->dOutSize, c->dFrameState, c->checksum ? c->dChecks : (const uint32_t*)nullptr, (uint32_t)(c->lzPredict != 0));
        HIPCHK(c, hipGetLastError());
    }
    if (c->lzAll) { }
    else if (hbmOnly)
        hipLaunchKernelGGL(zhip::k_frame_hbm, dim3((unsigned)nU), dim3(ZHIP_ENT_THREADS), lds, s,
                           (const uint8_t*)srcDev, c->dUnits, c->dSlots, (uint32_t)nU, c->dTabs, tabStride, c->dSeqs, c->dLits, c->dStBits,
                           c->dFrameOut, c->dOutSize, c->dFrameState, c->checksum ? c->dChecks : (const uint32_t*)nullptr, mt.on ? c->dJobs : (const zhip::ZhipJob*)nullptr);
    else
        hipLaunchKernelGGL(zhip::k_frame_fast, dim3((unsigned)nU), dim3(ZHIP_ENT_THREADS), lds, s,
                           (const uint8_t*)srcDev, c->dUnits, c->dSlots, (uint32_t)nU, c->dTabs, tabStride, c->dSeqs, c->dLits, c->dStBits,
                           c->dFrameOut, c->dOutSize, c->dFrameState, c->checksum ? c->dChecks : (const uint32_t*)nullptr, mt.on ? c->dJobs : (const zhip::ZhipJob*)nullptr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[2], s));
    hipLaunchKernelGGL(zhip::k_offsets, dim3(1), dim3(256), 0, s, c->dOutSize, (uint32_t)nU, c->dOutOff);
    hipLaunchKernelGGL(zhip::k_gather, dim3((unsigned)nU), dim3(256), 0, s, c->dFrameOut, c->dSlots, c->dOutSize, c->dOutOff, (uint32_t)nU, (uint8_t*)dstDev);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev[3], s));
    if (mt.on) {
        HIPCHK(c, hipMemsetAsync(c->dFrameSizes, 0, nFrames * sizeof(uint32_t), s));
        hipLaunchKernelGGL(zhip::k_frame_sizes, dim3((unsigned)((nU + 255) / 256)), dim3(256), 0, s, c->dOutSize, c->dJobs, (uint32_t)nU, c->dFrameSizes);
        if (frameSizesDev) HIPCHK(c, hipMemcpyAsync(frameSizesDev, c->dFrameSizes, nFrames * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    } else if (frameSizesDev) HIPCHK(c, hipMemcpyAsync(frameSizesDev, c->dOutSize, nFrames * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->hOutSize, c->dOutSize, nU * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    uint64_t total = 0;
    for (size_t i = 0; i < nU; i++) total += c->hOutSize[i];
    c->hFrameSizes.assign(nFrames, 0);
    for (size_t i = 0; i < nU; i++) c->hFrameSizes[mt.on ? c->hJobs[i].frameIdx : i] += c->hOutSize[i];
    read_timing(c);
    c->stats[0] = nU; c->stats[1] = totalSrc; c->stats[2] = total; c->stats[3] = 0; c->stats[4] = 0;
    c->nUnits = nU;
    return (size_t)total;
}

static size_t units_as_frames_locked(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, const unsigned long long* offs,
                                     size_t nFrames, int level, uint32_t* frameSizesDev, hipStream_t s)
{
    return frames_device_locked(c, dstDev, dstCapacity, srcDev, offs, nFrames, level, frameSizesDev, s);
}

size_t zhip_frames_bound(const unsigned long long* srcOffsets, size_t nFrames)
{
    size_t b = 0;
    for (size_t i = 0; i < nFrames; i++) b += zhip::host_compress_bound((size_t)(srcOffsets[i + 1] - srcOffsets[i]));
    return b;
}

size_t zhip_compress_frames_device(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, const unsigned long long* srcOffsets,
                                   size_t nFrames, int level, const unsigned cparams[7], uint32_t* frameSizesDev, void* stream)
{
    if (cparams && !zhip::host_check_overrides(cparams)) return ZERR(ZE_parameter_outOfBound);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    OvrScope scope(c, cparams);
    return frames_device_locked(c, dstDev, dstCapacity, srcDev, srcOffsets, nFrames, level, frameSizesDev, stream ? (hipStream_t)stream : c->stream);
}

// host buffers: stage the inputs on the device, run the frames (with or without jobs), copy the packed frames back
static size_t frames_host(zhip_ctx* c, void* dst, size_t dstCapacity, const void* src, const unsigned long long* srcOffsets,
                          size_t nFrames, int level, const unsigned cparams[7], size_t* frameSizes, MtParams mt)
{
    if (cparams && !zhip::host_check_overrides(cparams)) return ZERR(ZE_parameter_outOfBound);
    if (nFrames == 0) return ZERR(ZE_srcSize_wrong);
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    OvrScope scope(c, cparams);
    size_t const bound = zhip_frames_bound(srcOffsets, nFrames);
    if (dstCapacity < bound) return ZERR(ZE_dstSize_tooSmall);
    unsigned long long const lo = srcOffsets[0], hi = srcOffsets[nFrames];
    size_t const srcSize = (size_t)(hi - lo);
    if (c->srcStageCap < srcSize + 64) {
        (void)hipFree(c->dSrcStage); c->dSrcStage = nullptr; c->srcStageCap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dSrcStage, srcSize + 64)); c->srcStageCap = srcSize + 64;
    }
    if (c->dstStageCap < bound + 64) {
        (void)hipFree(c->dDstStage); c->dDstStage = nullptr; c->dstStageCap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dDstStage, bound + 64)); c->dstStageCap = bound + 64;
    }
    if (srcSize) HIPCHK(c, hipMemcpyAsync(c->dSrcStage, (const uint8_t*)src + lo, srcSize, hipMemcpyHostToDevice, c->stream));
    std::vector<unsigned long long> rel(nFrames + 1);
    for (size_t i = 0; i <= nFrames; i++) rel[i] = srcOffsets[i] - lo;
    size_t const total = frames_device_locked(c, c->dDstStage, c->dstStageCap, c->dSrcStage, rel.data(), nFrames, level, nullptr, c->stream, mt);
    if (zhip_isError(total)) return total;
    HIPCHK(c, hipMemcpy(dst, c->dDstStage, total, hipMemcpyDeviceToHost));
    if (frameSizes) for (size_t i = 0; i < nFrames; i++) frameSizes[i] = c->hFrameSizes[i];
    return total;
}

size_t zhip_compress_frames(zhip_ctx* c, void* dst, size_t dstCapacity, const void* src, const unsigned long long* srcOffsets,
                            size_t nFrames, int level, const unsigned cparams[7], size_t* frameSizes)
{
    return frames_host(c, dst, dstCapacity, src, srcOffsets, nFrames, level, cparams, frameSizes, MtParams{false, 0, 0});
}

// ZSTD_c_nbWorkers >= 1: the same inputs, each as the frame the reference's job pool produces
size_t zhip_compress_frames_mt_device(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, const unsigned long long* srcOffsets,
                                      size_t nFrames, int level, const unsigned cparams[7], size_t jobSize, int overlapLog, uint32_t* frameSizesDev, void* stream)
{
    if (cparams && !zhip::host_check_overrides(cparams)) return ZERR(ZE_parameter_outOfBound);
    if (overlapLog < 0 || overlapLog > 9) return ZERR(ZE_parameter_outOfBound);        // ZSTDMT_OVERLAPLOG_MIN / MAX
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    OvrScope scope(c, cparams);
    return frames_device_locked(c, dstDev, dstCapacity, srcDev, srcOffsets, nFrames, level, frameSizesDev, stream ? (hipStream_t)stream : c->stream, MtParams{true, jobSize, overlapLog});
}

size_t zhip_compress_frames_mt(zhip_ctx* c, void* dst, size_t dstCapacity, const void* src, const unsigned long long* srcOffsets,
                               size_t nFrames, int level, const unsigned cparams[7], size_t jobSize, int overlapLog, size_t* frameSizes)
{
    if (overlapLog < 0 || overlapLog > 9) return ZERR(ZE_parameter_outOfBound);
    return frames_host(c, dst, dstCapacity, src, srcOffsets, nFrames, level, cparams, frameSizes, MtParams{true, jobSize, overlapLog});
}

// One CHUNK of a job-pool frame on one context (zhip_compress_frame_mt_multi, zhip_multi.h): jobs[0..nJobs) in frame order, with
// their ABSOLUTE starts and prefix lengths; srcDev holds the frame's bytes from absolute offset w0 on (w0 = the first job's window
// start), so a unit's source offset is -w0 (mod 2^64: the kernel adds the absolute position back).  Packed output to dstDev.
static size_t frame_jobs_chunk_device(zhip_ctx* c, void* dstDev, size_t dstCapacity, const void* srcDev, unsigned long long w0,
                                      const zhip::CParams& cp, const zhip::ZhipJob* jobs, const uint32_t* lens, size_t nJobs, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (nJobs == 0 || nJobs > c->maxUnits) return ZERR(ZE_srcSize_wrong);
    if (nJobs * (size_t)ZHIP_SEQ_CAP > c->seqArena || nJobs * (size_t)ZHIP_LIT_STRIDE > c->litArena) return ZERR(ZE_srcSize_wrong);   // a records context: arenas too small for full-size slots
    size_t outBytes = 0, tabWords = 0, tabWordsAll = 0, bound = 0; uint32_t ldsTab = 0; unsigned long long total = 0;
    c->hJobs.assign(jobs, jobs + nJobs); c->hFrameUnits.resize(1);
    for (size_t i = 0; i < nJobs; i++) {
        size_t const len = lens[i];
        ZhipUnit& u = c->hUnits[i];
        u.srcOff = 0ull - w0; u.srcLen = (uint32_t)len;
        u.windowLog = (uint8_t)cp.windowLog; u.chainLog = (uint8_t)cp.chainLog; u.hashLog = (uint8_t)cp.hashLog;
        u.minMatch = (uint8_t)cp.minMatch; u.strategy = (uint8_t)cp.strategy; u.searchLog = (uint8_t)cp.searchLog;
        u.litMode = (cp.strategy == ZHIP_STRAT_FAST && cp.targetLength > 0) ? 1 : 0; u.pad0 = 0; u.targetLength = cp.targetLength; u.rowLog = 0; u.pad1 = 0;
        ZhipSlot& sl = c->hSlots[i];
        sl.seqOff = i * (uint64_t)ZHIP_SEQ_CAP; sl.litOff = i * (uint64_t)ZHIP_LIT_STRIDE; sl.outOff = outBytes; sl.seqCap = ZHIP_SEQ_CAP; sl.pad0 = 0;
        outBytes += (zhip::host_compress_bound(len) + 1024 + 15) & ~(size_t)15;
        bound += zhip::host_compress_bound(len);
        c->hJobs[i].frameIdx = 0;
        uint32_t const mode = zhip::frame_table_mode(cp.strategy, cp.hashLog, (unsigned long long)len + jobs[i].prefixLen + 1);
        size_t const w = zhip::frame_table_words(cp.strategy, cp.hashLog, cp.chainLog);
        if (w > tabWordsAll) tabWordsAll = w;
        if (mode == zhip::ZHIP_FT_HBM) { if (w > tabWords) tabWords = w; }
        else { uint32_t const b = zhip::frame_table_lds_bytes(mode, cp.hashLog); if (b > ldsTab) ldsTab = b; }
        total += len;
    }
    frames_prefer_hbm(c, nJobs, ldsTab, tabWords, tabWordsAll);
    if (dstCapacity < bound) return ZERR(ZE_dstSize_tooSmall);
    c->hFrameUnits[0] = c->hUnits[0];
    int const ck = c->checksum; c->checksum = 0;                          // a chunk does not see the whole frame: no checksum here
    size_t const r = frames_run_locked(c, dstDev, srcDev, nJobs, 1, ldsTab, tabWords, outBytes, total, true, nullptr, s);
    c->checksum = ck;
    return r;
}

// ------------------------------------------------------------------ seekable container
size_t zhip_seek_table_bound(size_t nFrames, int withChecksum) { return 8 + nFrames * (withChecksum ? 12 : 8) + 9; }

static inline void put32(uint8_t* p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

size_t zhip_write_seek_table(void* dstv, size_t cap, const unsigned* cSizes, const unsigned* dSizes, const unsigned* checksums, size_t nFrames)
{
    size_t const entry = checksums ? 12 : 8, total = 8 + nFrames * entry + 9;
    uint8_t* op = (uint8_t*)dstv;
    if (cap < total) return ZERR(ZE_dstSize_tooSmall);
    if (nFrames > 0x8000000u) return ZERR(ZE_parameter_outOfBound);          // ZSTD_SEEKABLE_MAXFRAMES (zstd_seekable.h:22)
    put32(op, 0x184D2A5Eu); put32(op + 4, (uint32_t)(total - 8)); op += 8;   // skippable frame header
    for (size_t i = 0; i < nFrames; i++) {
        put32(op, cSizes[i]); put32(op + 4, dSizes[i]); op += 8;
        if (checksums) { put32(op, checksums[i]); op += 4; }
    }
    put32(op, (uint32_t)nFrames); op[4] = (uint8_t)((checksums ? 1u : 0u) << 7); put32(op + 5, 0x8F92EAB1u);   // footer
    return total;
}

size_t zhip_compress_seekable(zhip_ctx* c, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, size_t unitSize)
{
    size_t const nUnitsMax = srcSize ? (srcSize + (unitSize ? unitSize : 1) - 1) / (unitSize ? unitSize : 1) : 1;
    if (dstCapacity < zhip_compressBound(srcSize, unitSize) + zhip_seek_table_bound(nUnitsMax, 1)) return ZERR(ZE_dstSize_tooSmall);
    size_t const total = zhip_compress(c, dst, dstCapacity, src, srcSize, level, unitSize, nullptr);
    if (zhip_isError(total)) return total;
    std::lock_guard<std::mutex> lk(c->mu);
    size_t const n = c->nUnits;
    std::vector<unsigned> cs(n), ds(n), ck;
    for (size_t i = 0; i < n; i++) { cs[i] = c->hOutSize[i]; ds[i] = c->hUnits[i].srcLen; }
    if (c->checksum) { ck.resize(n); HIPCHK(c, hipMemcpy(ck.data(), c->dChecks, n * sizeof(uint32_t), hipMemcpyDeviceToHost)); }
    size_t const t = zhip_write_seek_table((uint8_t*)dst + total, dstCapacity - total, cs.data(), ds.data(), c->checksum ? ck.data() : nullptr, n);
    if (zhip_isError(t)) return t;
    return total + t;
}

// ------------------------------------------------------------------ dictionary path (records)
struct zhip_cdict_s {
    zhip::HostCDict h;
    int device;
    uint8_t* dContent; uint32_t* dTabL; uint32_t* dTabS; zhip::ZhipDictEntropy* dEntropy;
};

void zhip_free_cdict(zhip_cdict* cd)
{
    if (!cd) return;
    (void)hipSetDevice(cd->device);
    (void)hipFree(cd->dContent); (void)hipFree(cd->dTabL); (void)hipFree(cd->dTabS); (void)hipFree(cd->dEntropy);
    delete cd;
}

zhip_cdict* zhip_create_cdict(int device, const void* dict, size_t dictSize, int level)
{
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    zhip_cdict* cd = new zhip_cdict_s();
    cd->device = device; cd->dContent = nullptr; cd->dTabL = nullptr; cd->dTabS = nullptr; cd->dEntropy = nullptr;
    if (zhip::host_cdict_build(cd->h, dict, dictSize, level) != 0) { delete cd; return nullptr; }
    bool ok = hipMalloc((void**)&cd->dContent, cd->h.content.size()) == hipSuccess;
    ok = ok && hipMalloc((void**)&cd->dTabL, cd->h.tabL.size() * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMalloc((void**)&cd->dTabS, cd->h.tabS.size() * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemcpy(cd->dContent, cd->h.content.data(), cd->h.content.size(), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(cd->dTabL, cd->h.tabL.data(), cd->h.tabL.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(cd->dTabS, cd->h.tabS.data(), cd->h.tabS.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    if (ok && cd->h.hasEntropy) {
        ok = hipMalloc((void**)&cd->dEntropy, sizeof(zhip::ZhipDictEntropy)) == hipSuccess;
        ok = ok && hipMemcpy(cd->dEntropy, &cd->h.ent, sizeof(zhip::ZhipDictEntropy), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok) { zhip_free_cdict(cd); return nullptr; }
    return cd;
}

size_t zhip_records_bound(const unsigned long long* recOffsets, size_t nRec)
{
    size_t b = 0;
    for (size_t i = 0; i < nRec; i++) b += zhip::host_compress_bound((size_t)(recOffsets[i + 1] - recOffsets[i]));
    return b;
}

static size_t compress_records_locked(zhip_ctx* c, const zhip_cdict* cd, void* dstDev, size_t dstCapacity, const void* srcDev,
                                      const unsigned long long* recOffsets, size_t nRec, uint32_t* frameSizesDev, hipStream_t s)
{
    if (nRec == 0) return 0;
    if (nRec > c->maxUnits) { snprintf(c->err, sizeof(c->err), "%zu records > context capacity %zu", nRec, c->maxUnits); return ZERR(ZE_srcSize_wrong); }
    // The per-record descriptors (parameters by size class, slots in the arenas) are filled by several host threads for large batches — 10 M records are
    // 640 MB of descriptors: each thread takes a contiguous range with offsets relative to its start, the ranges' totals are prefix-summed, a second
    // sweep adds the bases.  The destination bound of the call is summed in the same sweep.
    uint64_t seqOff = 0, litOff = 0, outOff = 0;
    uint32_t mhL = 6, mhS = 6, mhFast = 0; int fam = 0;
    bool const attached = cd->h.len != 0;                // dictionaries below 8 bytes are not attached, their parameters still apply
    std::vector<uint32_t> extIdx;                        // sources above the attach cut-off: the reference copies the dictionary (zstd_compress.c:2289-2315)
    struct Part { uint64_t seq = 0, lit = 0, out = 0, bound = 0; uint32_t mhL = 6, mhS = 6, mhFast = 0; int fam = 0; size_t bad = (size_t)-1; std::vector<uint32_t> ext; };
    unsigned nThreads = 1;
    if (nRec >= 262144) { unsigned const hw = std::thread::hardware_concurrency(); nThreads = hw >= 16 ? 16 : (hw ? hw : 1); }
    std::vector<Part> parts(nThreads);
    auto fill = [&](unsigned t) {
        Part& P = parts[t];
        size_t const a = nRec * t / nThreads, b = nRec * (t + 1) / nThreads;
        for (size_t i = a; i < b; i++) {
            size_t const n = (size_t)(recOffsets[i + 1] - recOffsets[i]);
            zhip::CParams cp;
            bool const copyParams = zhip::host_cdict_is_copy_mode(cd->h, n);     // the parameter rule also applies to an ignored (< 8 byte) dictionary
            bool const copyMode = attached && copyParams;
            if (n > ZHIP_UNIT_MAX || !(copyParams ? zhip::host_cdict_copy_params(cd->h, n, &cp) : zhip::host_cdict_unit_params(cd->h, n, &cp))) { P.bad = i; return; }
            ZhipUnit& u = c->hUnits[i];
            u.srcOff = recOffsets[i]; u.srcLen = (uint32_t)n;
            u.windowLog = (uint8_t)cp.windowLog; u.chainLog = (uint8_t)cp.chainLog; u.hashLog = (uint8_t)cp.hashLog;
            u.minMatch = (uint8_t)cp.minMatch; u.strategy = (uint8_t)cp.strategy; u.searchLog = (uint8_t)cp.searchLog;
            u.litMode = (cp.strategy == ZHIP_STRAT_FAST && cp.targetLength > 0) ? 1 : 0; u.pad0 = copyMode ? ZHIP_UNIT_COPYMODE : 0; u.targetLength = cp.targetLength; u.rowLog = 0; u.pad1 = 0;
            ZhipSlot& sl = c->hSlots[i];
            sl.seqOff = P.seq; sl.litOff = P.lit; sl.outOff = P.out; sl.seqCap = (uint32_t)rec_seq_cap(n); sl.pad0 = 0;
            P.seq += rec_seq_cap(n); P.lit += rec_lit_bytes(n); P.out += rec_out_bytes(n); P.bound += zhip::host_compress_bound(n);
            if (copyMode) { P.ext.push_back((uint32_t)i); continue; }            // its tables are the CDict's geometry, in HBM
            if (cp.strategy == ZHIP_STRAT_FAST) { P.fam |= 1; if (cp.hashLog > P.mhFast) P.mhFast = cp.hashLog; }
            else { P.fam |= 2; if (cp.hashLog > P.mhL) P.mhL = cp.hashLog; if (cp.chainLog > P.mhS) P.mhS = cp.chainLog; }
        }
    };
    auto rebase = [&](unsigned t, uint64_t bs, uint64_t bl, uint64_t bo) {
        size_t const a = nRec * t / nThreads, b = nRec * (t + 1) / nThreads;
        for (size_t i = a; i < b; i++) { ZhipSlot& sl = c->hSlots[i]; sl.seqOff += bs; sl.litOff += bl; sl.outOff += bo; }
    };
    // run f(t) for t in [first, nThreads): on threads where they can be had (thread creation may throw — rlimit, a container's pid cap — and nothing
    // may unwind through the C ABI), the rest on this one
    auto run_parts = [&](auto f, unsigned first) {
        std::vector<std::thread> th;
        unsigned t = first;
        if (nThreads - first > 1) {
            try { for (; t + 1 < nThreads; t++) th.emplace_back(f, t); } catch (...) { }
        }
        for (; t < nThreads; t++) f(t);
        for (auto& x : th) x.join();
    };
    run_parts(fill, 0);
    uint64_t bound = 0;
    std::vector<uint64_t> bs(nThreads), bl(nThreads), bo(nThreads);
    for (unsigned t = 0; t < nThreads; t++) {
        Part const& P = parts[t];
        if (P.bad != (size_t)-1) {
            snprintf(c->err, sizeof(c->err), "record %zu (%zu bytes): no parameter row (sources above 128 KB are not single-block frames)", P.bad, (size_t)(recOffsets[P.bad + 1] - recOffsets[P.bad]));
            return ZERR(ZE_parameter_unsupported);
        }
        bs[t] = seqOff; bl[t] = litOff; bo[t] = outOff;
        seqOff += P.seq; litOff += P.lit; outOff += P.out; bound += P.bound;
        fam |= P.fam; if (P.mhL > mhL) mhL = P.mhL; if (P.mhS > mhS) mhS = P.mhS; if (P.mhFast > mhFast) mhFast = P.mhFast;
        extIdx.insert(extIdx.end(), P.ext.begin(), P.ext.end());
    }
    if (dstCapacity < bound) return ZERR(ZE_dstSize_tooSmall);
    if (nThreads > 1) run_parts([&](unsigned t) { rebase(t, bs[t], bl[t], bo[t]); }, 1);
    if (seqOff > c->seqArena || litOff > c->litArena || outOff > c->outArena) {
        snprintf(c->err, sizeof(c->err), "records need %llu sequence slots / %llu literal bytes / %llu output bytes: context too small (zhip_create_for_records)",
                 (unsigned long long)seqOff, (unsigned long long)litOff, (unsigned long long)outOff);
        return ZERR(ZE_srcSize_wrong);
    }
    size_t r;
    if (!attached) {
        // no dictionary content: the ordinary kernels with the CDict-derived parameters
        c->strategy = fam; c->tabStride = 0; c->hcMaxLen = 0;
        if (fam & 2) {
            size_t w = zhip::dfast_table_bytes(mhL, mhS) >> 2; w = (w + 3) & ~(size_t)3;
            c->tabStride = w;
            if (c->tabsCap < nRec * w) {
                (void)hipFree(c->dTabs); c->dTabs = nullptr; c->tabsCap = 0;
                HIPCHK(c, hipMalloc((void**)&c->dTabs, nRec * w * sizeof(uint32_t))); c->tabsCap = nRec * w;
            }
        }
        r = launch_parse(c, (const uint8_t*)srcDev, nRec, mhFast, s);
    } else {
        zhip::ZhipCDictDev dv;
        dv.content = cd->dContent; dv.len = (uint32_t)cd->h.len; dv.hashLog = cd->h.cp.hashLog; dv.chainLog = cd->h.cp.chainLog;
        dv.minMatch = cd->h.cp.minMatch; dv.strategy = cd->h.cp.strategy; dv.tabL = cd->dTabL; dv.tabS = cd->dTabS;
        dv.rep[0] = cd->h.rep[0]; dv.rep[1] = cd->h.rep[1]; dv.rep[2] = cd->h.rep[2]; dv.dictID = cd->h.dictID;
        size_t const smem = (fam & 1) ? zhip::dict_fast_lds_bytes(mhFast) : zhip::dict_lds_bytes(mhL, mhS);
        HIPCHK(c, hipMemcpyAsync(c->dUnits, c->hUnits, nRec * sizeof(ZhipUnit), hipMemcpyHostToDevice, s));
        HIPCHK(c, hipMemcpyAsync(c->dSlots, c->hSlots, nRec * sizeof(ZhipSlot), hipMemcpyHostToDevice, s));
        if (smem > 64 * 1024) HIPCHK(c, hipFuncSetAttribute((const void*)zhip::k_parse_dict, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HIPCHK(c, hipEventRecord(c->ev[0], s));
        if (extIdx.size() < nRec) {
            // queue form (like the ZSTD_fast stage's): LDS-table wavefronts on this stream, global-table wavefronts beside them on coStream
            int perCU = 0;
            if (c->dictQueue && nRec > 1 &&
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, (const void*)zhip::k_parse_dict_q, 64, smem) == hipSuccess && perCU >= 1 &&
                (size_t)perCU * (size_t)c->numCUs * ZHIP_DICT_TICKET < nRec) {
                size_t const gridQ = (size_t)perCU * (size_t)c->numCUs;
                int gw = c->dictGWaves; if (gw > 32 - perCU) gw = 32 - perCU; if (gw < 0) gw = 0;
                size_t gridG = (size_t)gw * (size_t)c->numCUs;
                size_t const gtabFast = (fam & 1) ? (size_t)(2u << mhFast) : 0, gtabDfast = (fam & 2) ? (size_t)(2u << mhL) + (size_t)(2u << mhS) : 0;
                size_t const gtabBytes = ((gtabFast > gtabDfast ? gtabFast : gtabDfast) + 255) & ~(size_t)255;
                if (gridG && c->gtabsCap * sizeof(uint32_t) < gridG * gtabBytes) {
                    (void)hipFree(c->dGTabs); c->dGTabs = nullptr; c->gtabsCap = 0;
                    if (hipMalloc((void**)&c->dGTabs, gridG * gtabBytes) != hipSuccess) { (void)hipGetLastError(); gridG = 0; }
                    else c->gtabsCap = gridG * gtabBytes / sizeof(uint32_t);
                }
                HIPCHK(c, hipMemsetAsync(c->dQueue, 0, 64, s));
                if (gridG) HIPCHK(c, hipEventRecord(c->coEv[0], s));
                hipLaunchKernelGGL(zhip::k_parse_dict_q, dim3((unsigned)gridQ), dim3(64), smem, s,
                                   (const uint8_t*)srcDev, c->dUnits, c->dSlots, (uint32_t)nRec, dv, c->dSeqs, c->dLits, c->dParse, c->dQueue);
                if (gridG) {
                    HIPCHK(c, hipStreamWaitEvent(c->coStream, c->coEv[0], 0));
                    hipLaunchKernelGGL(zhip::k_parse_dict_g, dim3((unsigned)gridG), dim3(64), 0, c->coStream,
                                       (const uint8_t*)srcDev, c->dUnits, c->dSlots, (uint32_t)nRec, dv, c->dSeqs, c->dLits, c->dParse, c->dQueue,
                                       (unsigned char*)c->dGTabs, (uint32_t)gtabBytes);
                    HIPCHK(c, hipEventRecord(c->coEv[1], c->coStream));
                    HIPCHK(c, hipStreamWaitEvent(s, c->coEv[1], 0));
                }
            } else
            hipLaunchKernelGGL(zhip::k_parse_dict, dim3((unsigned)nRec), dim3(64), smem, s,
                               (const uint8_t*)srcDev, c->dUnits, c->dSlots, (uint32_t)nRec, dv, c->dSeqs, c->dLits, c->dParse);
        }
        HIPCHK(c, hipGetLastError());
        if (!extIdx.empty()) {                                  // copy mode: private table copies + one lane per source
            size_t const nExt = extIdx.size();
            size_t stride = zhip::ext_table_words(cd->h.cp.hashLog, cd->h.cp.chainLog, cd->h.cp.strategy); stride = (stride + 3) & ~(size_t)3;
            size_t const idxWords = (nExt + 3) & ~(size_t)3;
            if (c->tabsCap < nExt * stride + idxWords) {
                (void)hipFree(c->dTabs); c->dTabs = nullptr; c->tabsCap = 0;
                HIPCHK(c, hipMalloc((void**)&c->dTabs, (nExt * stride + idxWords) * sizeof(uint32_t))); c->tabsCap = nExt * stride + idxWords;
            }
            uint32_t* const dIdx = c->dTabs + nExt * stride;      // the index list rides behind the tables
            HIPCHK(c, hipMemcpyAsync(dIdx, extIdx.data(), nExt * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            uint32_t const wordsL = 1u << cd->h.cp.hashLog, wordsS = cd->h.cp.strategy == ZHIP_STRAT_DFAST ? 1u << cd->h.cp.chainLog : 0u;
            hipLaunchKernelGGL(zhip::k_ext_init, dim3(64, (unsigned)nExt), dim3(256), 0, s, cd->dTabL, cd->dTabS, wordsL, wordsS, c->dTabs, stride);
            hipLaunchKernelGGL(zhip::k_parse_ext, dim3((unsigned)((nExt + 63) / 64)), dim3(64), 0, s,
                               (const uint8_t*)srcDev, c->dUnits, c->dSlots, dIdx, (uint32_t)nExt, dv, c->dTabs, stride, c->dSeqs, c->dLits, c->dParse);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(c->ev[1], s));               // the match-finder time of the call includes them
            HIPCHK(c, hipStreamSynchronize(s));                   // extIdx (host vector) must outlive the copy
        }
        HIPCHK(c, hipEventRecord(c->ev[1], s));
        c->hcEvUsed = 0;
        r = 0;
    }
    if (zhip_isError(r)) return r;
    c->curDictEntropy = cd->dEntropy; c->curDictID = cd->h.dictID;           // the first block of every frame starts from the dictionary's entropy state
    r = launch_entropy_gather(c, (const uint8_t*)srcDev, nRec, (uint8_t*)dstDev, s);
    c->curDictEntropy = nullptr; c->curDictID = 0;
    if (zhip_isError(r)) return r;
    if (frameSizesDev) HIPCHK(c, hipMemcpyAsync(frameSizesDev, c->dOutSize, nRec * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->hOutSize, c->dOutSize, nRec * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    uint64_t total = 0;
    for (size_t i = 0; i < nRec; i++) total += c->hOutSize[i];
    read_timing(c);
    c->stats[0] = nRec; c->stats[1] = (unsigned long long)(recOffsets[nRec] - recOffsets[0]); c->stats[2] = total; c->stats[3] = 0; c->stats[4] = 0;
    c->nUnits = nRec;
    return (size_t)total;
}

size_t zhip_compress_records_device(zhip_ctx* c, const zhip_cdict* cd, void* dstDev, size_t dstCapacity, const void* srcDev,
                                    const unsigned long long* recOffsets, size_t nRec, uint32_t* frameSizesDev, void* stream)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!cd || cd->device != c->device) { snprintf(c->err, sizeof(c->err), "dictionary belongs to another device"); return ZERR(ZE_GENERIC); }
    return compress_records_locked(c, cd, dstDev, dstCapacity, srcDev, recOffsets, nRec, frameSizesDev, stream ? (hipStream_t)stream : c->stream);
}

size_t zhip_compress_records(zhip_ctx* c, const zhip_cdict* cd, void* dst, size_t dstCapacity, const void* src,
                             const unsigned long long* recOffsets, size_t nRec, size_t* frameSizes)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (!cd || cd->device != c->device) { snprintf(c->err, sizeof(c->err), "dictionary belongs to another device"); return ZERR(ZE_GENERIC); }
    if (nRec == 0) return 0;
    size_t const bound = zhip_records_bound(recOffsets, nRec), srcSize = (size_t)recOffsets[nRec];
    if (dstCapacity < bound) return ZERR(ZE_dstSize_tooSmall);
    if (c->srcStageCap < srcSize + 64) {
        (void)hipFree(c->dSrcStage); c->dSrcStage = nullptr; c->srcStageCap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dSrcStage, srcSize + 64)); c->srcStageCap = srcSize + 64;
    }
    if (c->dstStageCap < bound + 64) {
        (void)hipFree(c->dDstStage); c->dDstStage = nullptr; c->dstStageCap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dDstStage, bound + 64)); c->dstStageCap = bound + 64;
    }
    if (srcSize) HIPCHK(c, hipMemcpyAsync(c->dSrcStage, src, srcSize, hipMemcpyHostToDevice, c->stream));
    size_t const total = compress_records_locked(c, cd, c->dDstStage, c->dstStageCap, c->dSrcStage, recOffsets, nRec, nullptr, c->stream);
    if (zhip_isError(total)) return total;
    HIPCHK(c, hipMemcpy(dst, c->dDstStage, total, hipMemcpyDeviceToHost));
    if (frameSizes) for (size_t i = 0; i < nRec; i++) frameSizes[i] = c->hOutSize[i];
    return total;
}

size_t zhip_parse_device(zhip_ctx* c, const void* srcDev, size_t srcSize, int level, size_t unitSize, void* stream)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    size_t err = 0; uint32_t mh = 0;
    size_t const nUnits = build_units(c, srcSize, unitSize, level, &err, &mh);
    if (!nUnits) return err;
    size_t const r = launch_parse(c, (const uint8_t*)srcDev, nUnits, mh, s);
    if (zhip_isError(r)) return r;
    HIPCHK(c, hipMemcpyAsync(c->hParse, c->dParse, nUnits * sizeof(ZhipParse), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    float ms = 0; HIPCHK(c, hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->timing[0] = ms; c->timing[1] = c->timing[2] = 0; c->timing[3] = ms;
    c->nUnits = nUnits;
    c->stats[0] = nUnits; c->stats[1] = srcSize; c->stats[2] = 0; c->stats[3] = 0; c->stats[4] = 0;
    return nUnits;
}

// device seqDef records -> ZSTD_Sequence[] (lib/compress/zstd_compress.c:3371-3454)
static size_t seqs_to_public(const ZhipSeq* s, const ZhipParse& m, zhip_Sequence* out, size_t cap)
{
    if ((size_t)m.nbSeq + 1 > cap) return ZERR(ZE_dstSize_tooSmall);
    uint32_t rep[3] = {1, 4, 8};
    for (uint32_t i = 0; i < m.nbSeq; i++) {
        uint32_t ll = s[i].litLength, ml = (uint32_t)s[i].mlBase + 3, ob = s[i].offBase, raw, rf = 0;
        if (m.longType == 1 && i == m.longPos) ll += 0x10000;
        if (m.longType == 2 && i == m.longPos) ml += 0x10000;
        if (ob <= 3) { rf = ob; raw = ll ? rep[ob - 1] : (ob == 3 ? rep[0] - 1 : rep[ob]); }
        else raw = ob - 3;
        out[i].offset = raw; out[i].litLength = ll; out[i].matchLength = ml; out[i].rep = rf;
        if (ob > 3) { rep[2] = rep[1]; rep[1] = rep[0]; rep[0] = ob - 3; }
        else { uint32_t const rc = ob - 1 + (ll == 0);
               if (rc) { uint32_t const cur = rc == 3 ? rep[0] - 1 : rep[rc]; rep[2] = rc >= 2 ? rep[1] : rep[2]; rep[1] = rep[0]; rep[0] = cur; } }
    }
    out[m.nbSeq].offset = 0; out[m.nbSeq].litLength = m.lastLits; out[m.nbSeq].matchLength = 0; out[m.nbSeq].rep = 0;
    return (size_t)m.nbSeq + 1;
}

// stats[0] units, [1] source bytes, [2] compressed bytes, [3] sequences, [4] literal bytes (fetched from the device on demand)
size_t zhip_last_stats(zhip_ctx* c, unsigned long long stats[5])
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->nUnits) {
        HIPCHK(c, hipMemcpy(c->hParse, c->dParse, c->nUnits * sizeof(ZhipParse), hipMemcpyDeviceToHost));
        unsigned long long ns = 0, nl = 0;
        for (size_t i = 0; i < c->nUnits; i++) { ns += c->hParse[i].nbSeq; nl += c->hParse[i].litSize; }
        c->stats[3] = ns; c->stats[4] = nl;
    }
    for (int i = 0; i < 5; i++) stats[i] = c->stats[i];
    return 0;
}

size_t zhip_get_sequences(zhip_ctx* c, size_t unitIndex, zhip_Sequence* out, size_t capacity)
{
    std::lock_guard<std::mutex> lk(c->mu);
    if (unitIndex >= c->nUnits) return ZERR(ZE_parameter_outOfBound);
    HIPCHK(c, hipSetDevice(c->device));
    ZhipParse const m = c->hParse[unitIndex];
    std::vector<ZhipSeq> tmp(m.nbSeq ? m.nbSeq : 1);
    if (m.nbSeq) HIPCHK(c, hipMemcpy(tmp.data(), c->dSeqs + c->hSlots[unitIndex].seqOff, m.nbSeq * sizeof(ZhipSeq), hipMemcpyDeviceToHost));
    return seqs_to_public(tmp.data(), m, out, capacity);
}

}  // extern "C"

#ifdef ZHIP_PROF
// measurement variant only: sums of per-phase s_memtime ticks over all workgroups since the last reset
extern "C" void zhip_prof_read(unsigned long long out[32], int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(zhip::g_prof), 32 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[32]; memset(z, 0, sizeof(z)); (void)hipMemcpyToSymbol(HIP_SYMBOL(zhip::g_prof), z, sizeof(z)); }
}
// the ZSTD_fast window's phases (zhip_parse.h, WPH_*): [0,32) ticks, [32,64) visits
extern "C" void zhip_wph_read(unsigned long long out[64], int reset)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(zhip::g_wph), 64 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[64]; memset(z, 0, sizeof(z)); (void)hipMemcpyToSymbol(HIP_SYMBOL(zhip::g_wph), z, sizeof(z)); }
}
#endif

// ------------------------------------------------------------------ block-level plugin (B1)
// content fingerprint of a prepared block (two independent multiply-xor lanes over 8-byte words): a cached parse is only
// served when the bytes at that address are still the bytes that were parsed — callers refill and reuse buffers
static void block_fingerprint(const uint8_t* p, size_t n, uint64_t out[2])
{
    uint64_t a = 0x9E3779B97F4A7C15ull ^ n, b = 0xC2B2AE3D27D4EB4Full + n;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w; memcpy(&w, p + i, 8);
        a = (a ^ w) * 0xFF51AFD7ED558CCDull; a ^= a >> 32;
        b = (b + w) * 0xC4CEB9FE1A85EC53ull; b ^= b >> 29;
    }
    uint64_t w = 0; memcpy(&w, p + i, n - i);
    a = (a ^ w) * 0xFF51AFD7ED558CCDull; a ^= a >> 33;
    b = (b + w) * 0xC4CEB9FE1A85EC53ull; b ^= b >> 31;
    out[0] = a; out[1] = b;
}

// parse `srcSize` host bytes cut into blockSize blocks (each without history); results copied to the host cache
// fingerprints of the blocks [lo, hi) of a prepared buffer
static void fingerprint_range(const uint8_t* src, const ZhipUnit* units, size_t lo, size_t hi, uint64_t* out)
{
    for (size_t i = lo; i < hi; i++) block_fingerprint(src + units[i].srcOff, units[i].srcLen, out + 2 * i);
}

// Parse every block of the buffer in one launch and bring the sequences to the host: H2D, match finder, ONE packed copy back
// (k_seq_compact), while host threads fingerprint the blocks.  With `local` == nullptr the prepared cache is replaced under cacheMu at the very
// end (zhip_prepare_sequences); with `local` the result goes there and the shared cache is never touched (the producer's one-shot parse of an
// unprepared block: callbacks of other threads keep finding their prepared blocks while it runs — round-5 advisor finding).
struct PreparedLocal { std::vector<ZhipSeq> seqs; std::vector<ZhipParse> parse; };
static size_t prepare_locked(zhip_ctx* c, const void* src, size_t srcSize, size_t blockSize, int level, PreparedLocal* local = nullptr)
{
    if (blockSize == 0 || blockSize > ZHIP_UNIT_MAX) blockSize = ZHIP_UNIT_MAX;
    size_t err = 0; uint32_t mh = 0;
    size_t const nUnits = build_units(c, srcSize, blockSize, level, &err, &mh);
    if (!nUnits) return err;
    if (c->srcStageCap < srcSize + 64) {
        (void)hipFree(c->dSrcStage); c->dSrcStage = nullptr; c->srcStageCap = 0;
        HIPCHK(c, hipMalloc((void**)&c->dSrcStage, srcSize + 64)); c->srcStageCap = srcSize + 64;
    }
    if (srcSize) HIPCHK(c, hipMemcpyAsync(c->dSrcStage, src, srcSize, hipMemcpyHostToDevice, c->stream));
    size_t const r = launch_parse(c, c->dSrcStage, nUnits, mh, c->stream);
    if (zhip_isError(r)) return r;
    HIPCHK(c, hipMemcpyAsync(c->hParse, c->dParse, nUnits * sizeof(ZhipParse), hipMemcpyDeviceToHost, c->stream));
    // the content fingerprints (what lets a later callback trust the cache) while the device works: up to 16 host threads for large buffers
    std::vector<uint64_t> hashes(local ? 0 : 2 * nUnits);
    std::vector<std::thread> workers;
    size_t nThr = srcSize >= ((size_t)32 << 20) ? 16 : 1;
    if (nThr > nUnits) nThr = nUnits;
    if (local) nThr = 0;                                                           // a one-shot parse is never looked up by content
    if (nThr > 1) {
        try {
            for (size_t t = 1; t < nThr; t++)
                workers.emplace_back(fingerprint_range, (const uint8_t*)src, (const ZhipUnit*)c->hUnits, nUnits * t / nThr, nUnits * (t + 1) / nThr, hashes.data());
        } catch (...) { for (auto& w : workers) w.join(); workers.clear(); nThr = 1; }      // no threads to be had: this one does it all
    }
    if (!local) fingerprint_range((const uint8_t*)src, c->hUnits, 0, nThr > 1 ? nUnits / nThr : nUnits, hashes.data());
    for (auto& w : workers) w.join();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {   float ms = 0; (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);           // the match finder's own duration (zhip_last_timing: parse_ms)
        c->timing[0] = ms; c->timing[1] = c->timing[2] = 0; c->timing[3] = ms; }
    c->stats[0] = nUnits; c->stats[1] = srcSize; c->stats[2] = 0; c->stats[3] = 0; c->stats[4] = 0;
    std::vector<uint64_t> seqOff(nUnits + 1);
    size_t totalSeq = 0;
    for (size_t i = 0; i < nUnits; i++) { seqOff[i] = totalSeq; totalSeq += c->hParse[i].nbSeq; }
    seqOff[nUnits] = totalSeq;
    std::vector<ZhipSeq> seqs(totalSeq ? totalSeq : 1);
    if (totalSeq) {
        if (c->seqPackCap < totalSeq) {
            (void)hipFree(c->dSeqPack); c->dSeqPack = nullptr; c->seqPackCap = 0;
            HIPCHK(c, hipMalloc((void**)&c->dSeqPack, totalSeq * sizeof(ZhipSeq))); c->seqPackCap = totalSeq;
        }
        if (c->seqPackOffCap < nUnits) {
            (void)hipFree(c->dSeqPackOff); c->dSeqPackOff = nullptr; c->seqPackOffCap = 0;
            HIPCHK(c, hipMalloc((void**)&c->dSeqPackOff, nUnits * sizeof(uint64_t))); c->seqPackOffCap = nUnits;
        }
        HIPCHK(c, hipMemcpyAsync(c->dSeqPackOff, seqOff.data(), nUnits * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(zhip::k_seq_compact, dim3((unsigned)nUnits), dim3(256), 0, c->stream, c->dSeqs, c->dSlots, c->dParse, c->dSeqPackOff, (uint32_t)nUnits, c->dSeqPack);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(seqs.data(), c->dSeqPack, totalSeq * sizeof(ZhipSeq), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (local) { local->seqs.swap(seqs); local->parse.assign(c->hParse, c->hParse + nUnits); c->nUnits = nUnits; return nUnits; }
    {   std::lock_guard<std::mutex> lk(c->cacheMu);
        c->cacheParse.assign(c->hParse, c->hParse + nUnits);
        c->cacheUnits.assign(c->hUnits, c->hUnits + nUnits);
        c->cacheSeqs.swap(seqs); c->cacheSeqOff.swap(seqOff); c->cacheHash.swap(hashes);
        c->cacheSrc = src; c->cacheSize = srcSize; c->cacheBlock = blockSize; c->cacheLevel = level;
    }
    c->nUnits = nUnits;
    return nUnits;
}

extern "C" {

size_t zhip_prepare_sequences(zhip_ctx* c, const void* src, size_t srcSize, size_t blockSize, int level)
{
    std::lock_guard<std::mutex> lk(c->mu);
    HIPCHK(c, hipSetDevice(c->device));
    return prepare_locked(c, src, srcSize, blockSize, level);
}

// ZSTD_sequenceProducer_F (lib/zstd.h:2838).  Any failure -> ZHIP_SEQUENCE_PRODUCER_ERROR (so the caller's
// ZSTD_c_enableSeqProducerFallback decides what happens, lib/compress/zstd_compress.c:3338-3356).
size_t zhip_sequence_producer(void* state, zhip_Sequence* outSeqs, size_t outSeqsCapacity,
                              const void* src, size_t srcSize, const void* dict, size_t dictSize,
                              int compressionLevel, size_t windowSize)
{
    zhip_ctx* c = (zhip_ctx*)state;
    (void)windowSize;
    if (!c || dict != nullptr || dictSize != 0 || srcSize > ZHIP_UNIT_MAX || srcSize == 0) return ZHIP_SEQUENCE_PRODUCER_ERROR;
    const uint8_t* p = (const uint8_t*)src;
    // a prepared block: served from the cache without touching the device or the context's own lock, so the callbacks of many CCtx on many
    // host threads (one zhip_prepare_sequences, N x ZSTD_compress2) only meet in cacheMu for the copy.  The same address range is not
    // enough: the bytes must be the ones that were parsed (fingerprint taken before the lock)
    uint64_t fp[2]; block_fingerprint(p, srcSize, fp);
    {   std::lock_guard<std::mutex> lk(c->cacheMu);
        const uint8_t* base = (const uint8_t*)c->cacheSrc;
        bool hit = base && c->cacheLevel == compressionLevel && p >= base && p + srcSize <= base + c->cacheSize
                   && ((size_t)(p - base) % c->cacheBlock) == 0;
        size_t const idx = hit ? (size_t)(p - base) / c->cacheBlock : 0;
        if (hit && c->cacheUnits[idx].srcLen != srcSize) hit = false;
        if (hit && (fp[0] != c->cacheHash[2 * idx] || fp[1] != c->cacheHash[2 * idx + 1])) hit = false;
        if (hit) {
            size_t const r = seqs_to_public(c->cacheSeqs.data() + c->cacheSeqOff[idx], c->cacheParse[idx], outSeqs, outSeqsCapacity);
            return zhip_isError(r) ? ZHIP_SEQUENCE_PRODUCER_ERROR : r;
        }
    }
    // not prepared: one launch for this block (latency-bound path).  Its result stays local: the cache a zhip_prepare_sequences call
    // installed is neither replaced nor hidden while this runs: blocks of other sizes (the reference splits some 128 KB blocks at 92 KB,
    // lib/compress/zstd_compress.c:4494-4518) must not cost the prepared blocks their parse
    std::lock_guard<std::mutex> lk(c->mu);
    if (hipSetDevice(c->device) != hipSuccess) return ZHIP_SEQUENCE_PRODUCER_ERROR;
    PreparedLocal one;
    size_t const rp = prepare_locked(c, src, srcSize, srcSize, compressionLevel, &one);
    size_t r = rp;
    if (!zhip_isError(rp)) r = seqs_to_public(one.seqs.data(), one.parse[0], outSeqs, outSeqsCapacity);
    return zhip_isError(r) ? ZHIP_SEQUENCE_PRODUCER_ERROR : r;
}

}  // extern "C"

#include "zhip_multi.h"      // zhip_compress_multi: host buffers over several devices, pinned double-buffered lanes, ordered gather
#include "zhip_declib.h"     // decoder entry points (zhip_create_dctx, zhip_decompress, ...)
