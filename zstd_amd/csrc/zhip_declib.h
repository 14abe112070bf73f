// zhip_declib.h — host side of the decoder entry points of libzstd_hip.so (include/zstd_hip.h, "decompression"); included
// at the end of zhip_lib.hip (one translation unit owns the kernels).  The per-frame work is k_decode (zhip_decode.h); this
// file walks frame headers, sizes the scratch, launches and collects.
#pragma once
#include "zhip_ddict_host.h"

#define DERR(c) ((size_t)-(long)(c))

struct zhip_ddict_s {
    int device;
    zhip::HostDDict h;
    uint8_t* dContent; uint16_t* dHuf; uint32_t* dHuf2; uint64_t* dFse;
    ZhipDDictDev dev;
};

struct zhip_dctx_s {
    int device; int nCU;
    hipStream_t stream; hipEvent_t ev[4];
    uint32_t grid;                             // resident workgroups = scratch slots
    uint8_t* dLit; ZhipDSeq* dRecs; uint32_t* dCounter; uint64_t* dDefTabs;
    ZhipDFrame* dFrames; ZhipDResult* dResults; ZhipUnit* dUnits; uint32_t* dChecks; size_t framesCap;
    ZhipDFrame* hFrames; ZhipDResult* hResults; ZhipUnit* hUnits;
    uint8_t* dSrcStage; size_t srcStageCap; uint8_t* dDstStage; size_t dstStageCap;
    // one large frame, block-parallel (zhip_decode_big.h): block table, literal / record arenas, the copy map; grown on demand
    ZhipBfBlock* dBfBlocks = nullptr; ZhipBfInfo* dBfInfo = nullptr; uint8_t* dBfLit = nullptr; ZhipDSeq* dBfRecs = nullptr; uint32_t* dBfMap = nullptr;
    size_t bfBlocksCap = 0, bfLitCap = 0, bfRecsCap = 0, bfMapCap = 0;
    hipEvent_t bfEv[2] = { nullptr, nullptr };
    unsigned long long bigMin = 8ull << 20;        // frames stating at least this much content take the block-parallel path (zhip_dctx_set_bigframe_min, 0 = never)
    unsigned bfLast[4] = { 0, 0, 0, 0 };           // last call: frames decoded block-parallel, frames that fell back, jump rounds, blocks
    const uint8_t* bfHostSrc = nullptr;            // set by zhip_decompress for the duration of a call: the frames also lie in host memory at this address (same offsets as in the staged copy)
    std::vector<ZhipBfBlock> bfHostBlocks;
    double timing[2];
    std::mutex mu;
    char err[256];
};

#define DCHK(ctx, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    snprintf((ctx)->err, sizeof((ctx)->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    return DERR(1); } } while (0)

extern "C" {

void zhip_free_ddict(zhip_ddict* d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipFree(d->dContent); (void)hipFree(d->dHuf); (void)hipFree(d->dHuf2); (void)hipFree(d->dFse);
    delete d;
}

zhip_ddict* zhip_create_ddict(int device, const void* dict, size_t dictSize)
{
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    zhip_ddict* d = new zhip_ddict_s();
    d->device = device; d->dContent = nullptr; d->dHuf = nullptr; d->dHuf2 = nullptr; d->dFse = nullptr;
    if (zhip::host_ddict_build(d->h, dict, dictSize) != 0) { delete d; return nullptr; }
    size_t const n = d->h.content.size();
    bool ok = hipMalloc((void**)&d->dContent, n + 16) == hipSuccess;
    ok = ok && hipMalloc((void**)&d->dHuf, 4096 * sizeof(uint16_t)) == hipSuccess;
    ok = ok && hipMalloc((void**)&d->dFse, 1280 * sizeof(uint64_t)) == hipSuccess;
    ok = ok && hipMalloc((void**)&d->dHuf2, 2048 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipMemcpy(d->dHuf2, d->h.huf2.data(), 2048 * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && (n == 0 || hipMemcpy(d->dContent, d->h.content.data(), n, hipMemcpyHostToDevice) == hipSuccess);
    ok = ok && hipMemcpy(d->dHuf, d->h.huf.data(), 4096 * sizeof(uint16_t), hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(d->dFse, d->h.fse.data(), 1280 * sizeof(uint64_t), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { zhip_free_ddict(d); return nullptr; }
    memset(&d->dev, 0, sizeof(d->dev));
    d->dev.content = d->dContent; d->dev.len = (uint32_t)n; d->dev.dictID = d->h.dictID; d->dev.hasEntropy = d->h.hasEntropy; d->dev.hufLog = d->h.hufLog;
    d->dev.huf = d->dHuf; d->dev.huf2 = d->dHuf2; d->dev.fse = d->dFse;
    for (int k = 0; k < 3; k++) { d->dev.log[k] = d->h.log[k]; d->dev.rep[k] = d->h.rep[k]; }
    return d;
}

unsigned zhip_ddict_id(const zhip_ddict* d) { return d ? d->h.dictID : 0; }

void zhip_free_dctx(zhip_dctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->dLit); (void)hipFree(c->dRecs); (void)hipFree(c->dCounter); (void)hipFree(c->dDefTabs);
    (void)hipFree(c->dFrames); (void)hipFree(c->dResults); (void)hipFree(c->dUnits); (void)hipFree(c->dChecks);
    (void)hipFree(c->dSrcStage); (void)hipFree(c->dDstStage);
    (void)hipFree(c->dBfBlocks); (void)hipFree(c->dBfInfo); (void)hipFree(c->dBfLit); (void)hipFree(c->dBfRecs); (void)hipFree(c->dBfMap);
    for (int i = 0; i < 2; i++) if (c->bfEv[i]) (void)hipEventDestroy(c->bfEv[i]);
    (void)hipHostFree(c->hFrames); (void)hipHostFree(c->hResults); (void)hipHostFree(c->hUnits);
    for (int i = 0; i < 4; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

zhip_dctx* zhip_create_dctx(int device)
{
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    zhip_dctx* c = new zhip_dctx_s();
    c->device = device; c->stream = nullptr; c->err[0] = 0; c->timing[0] = c->timing[1] = 0;
    c->dLit = nullptr; c->dRecs = nullptr; c->dCounter = nullptr; c->dDefTabs = nullptr; c->dFrames = nullptr; c->dResults = nullptr; c->dUnits = nullptr; c->dChecks = nullptr;
    c->hFrames = nullptr; c->hResults = nullptr; c->hUnits = nullptr; c->framesCap = 0;
    c->dSrcStage = nullptr; c->srcStageCap = 0; c->dDstStage = nullptr; c->dstStageCap = 0;
    for (int i = 0; i < 4; i++) c->ev[i] = nullptr;
    hipDeviceProp_t prop;
    bool ok = hipGetDeviceProperties(&prop, device) == hipSuccess;
    c->nCU = ok ? prop.multiProcessorCount : 256;
    // resident workgroups per CU: what registers and LDS (sizeof(DecShared) of 160 KB) allow
    int perCU = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, zhip::k_decode, ZHIP_DEC_THREADS, sizeof(zhip::DecShared)) != hipSuccess || perCU < 1) perCU = 4;
    c->grid = (uint32_t)(c->nCU * perCU);
    ok = ok && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 4 && ok; i++) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dLit, (size_t)c->grid * ZHIP_DEC_LIT_STRIDE) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dRecs, (size_t)c->grid * 2 * (ZHIP_DEC_CHUNK + 1) * sizeof(ZhipDSeq)) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dCounter, 64) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->dDefTabs, 160 * sizeof(uint64_t)) == hipSuccess;
    if (ok) { uint64_t t[160]; zhip::host_dec_default_tables(t); ok = hipMemcpy(c->dDefTabs, t, sizeof(t), hipMemcpyHostToDevice) == hipSuccess; }
    if (!ok) { zhip_free_dctx(c); return nullptr; }
    return c;
}

// block-parallel path: minimum stated content size (0 = never), and what the last call did with it:
// out[0] frames decoded block-parallel, [1] frames that fell back to k_decode, [2] pointer-jumping rounds, [3] blocks
void zhip_dctx_set_bigframe_min(zhip_dctx* c, unsigned long long minContent) { if (c) c->bigMin = minContent; }
void zhip_dctx_last_bigframe(const zhip_dctx* c, unsigned out[4]) { for (int i = 0; i < 4; i++) out[i] = c ? c->bfLast[i] : 0; }

const char* zhip_dctx_last_error(const zhip_dctx* c) { return c ? c->err : "null context"; }
void zhip_dctx_last_timing(const zhip_dctx* c, double t[2]) { t[0] = c->timing[0]; t[1] = c->timing[1]; }

// = ZSTD_findFrameCompressedSize (lib/zstd.h:214): the frame (or skippable frame) at the start of src, which may hold more
size_t zhip_frame_compressed_size(const void* srcv, size_t srcSize)
{
    const uint8_t* const src = (const uint8_t*)srcv;
    if (srcSize >= 8) { uint32_t magic, sk; memcpy(&magic, src, 4); memcpy(&sk, src + 4, 4); if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) return (size_t)sk + 8 <= srcSize ? (size_t)sk + 8 : DERR(72); }
    size_t cs; uint64_t content, bound;
    int const e = zhip::host_frame_extent(src, srcSize, &cs, &content, &bound);
    return e ? DERR(e) : cs;
}

// = ZSTD_findFrameCompressedSize + ZSTD_getFrameContentSize over concatenated frames (lib/zstd.h:214, :196), skippable
// frames (magic 0x184D2A5?) are stepped over like ZSTD_decompress does (zstd_decompress.c:1100-1110).
size_t zhip_find_frames(const void* srcv, size_t srcSize, unsigned long long* srcOffsets, unsigned long long* srcSizes,
                        unsigned long long* contentSizes, unsigned long long* contentBounds, size_t maxFrames)
{
    const uint8_t* const src = (const uint8_t*)srcv; size_t pos = 0, n = 0;
    while (srcSize - pos >= 5) {
        uint32_t magic; memcpy(&magic, src + pos, 4);
        if (srcSize - pos >= 8 && (magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            uint32_t sk; memcpy(&sk, src + pos + 4, 4);
            if ((size_t)sk + 8 > srcSize - pos) return DERR(72);
            pos += (size_t)sk + 8; continue;
        }
        size_t cs; uint64_t content, bound;
        int const e = zhip::host_frame_extent(src + pos, srcSize - pos, &cs, &content, &bound);
        if (e) return DERR(e);
        if (n < maxFrames) {
            if (srcOffsets) srcOffsets[n] = pos;
            if (srcSizes) srcSizes[n] = cs;
            if (contentSizes) contentSizes[n] = content;
            if (contentBounds) contentBounds[n] = bound;
        }
        n++; pos += cs;
    }
    if (pos != srcSize) return DERR(72);                        // zstd_decompress.c:1195 "input not entirely consumed"
    return n;
}

}  // extern "C"

static size_t ensure_frames(zhip_dctx* c, size_t n)
{
    if (c->framesCap >= n) return 0;
    size_t const cap = n + n / 2 + 64;
    (void)hipFree(c->dFrames); (void)hipFree(c->dResults); (void)hipFree(c->dUnits); (void)hipFree(c->dChecks);
    (void)hipHostFree(c->hFrames); (void)hipHostFree(c->hResults); (void)hipHostFree(c->hUnits);
    c->dFrames = nullptr; c->dResults = nullptr; c->dUnits = nullptr; c->dChecks = nullptr; c->hFrames = nullptr; c->hResults = nullptr; c->hUnits = nullptr; c->framesCap = 0;
    DCHK(c, hipMalloc((void**)&c->dFrames, cap * sizeof(ZhipDFrame)));
    DCHK(c, hipMalloc((void**)&c->dResults, cap * sizeof(ZhipDResult)));
    DCHK(c, hipMalloc((void**)&c->dUnits, cap * sizeof(ZhipUnit)));
    DCHK(c, hipMalloc((void**)&c->dChecks, (cap + 16) * sizeof(uint32_t)));
    DCHK(c, hipHostMalloc((void**)&c->hFrames, cap * sizeof(ZhipDFrame), hipHostMallocDefault));
    DCHK(c, hipHostMalloc((void**)&c->hResults, cap * sizeof(ZhipDResult), hipHostMallocDefault));
    DCHK(c, hipHostMalloc((void**)&c->hUnits, cap * sizeof(ZhipUnit), hipHostMallocDefault));
    c->framesCap = cap;
    return 0;
}

// One frame through the block-parallel decoder (zhip_decode_big.h).  true: res is its result (status 0).  false: not a frame for this
// path, or the path met anything it does not handle — the caller leaves the frame to k_decode, which reports the reference's codes.
template <typename T> static bool bf_grow(T*& p, size_t& cap, size_t need)
{
    if (cap >= need) return true;
    (void)hipFree(p); p = nullptr; cap = 0;
    if (hipMalloc((void**)&p, need * sizeof(T)) != hipSuccess) { (void)hipGetLastError(); return false; }
    cap = need; return true;
}
static bool bigframe_decode(zhip_dctx* c, const ZhipDFrame& f, uint8_t* dstDev, const uint8_t* srcDev, hipStream_t s, ZhipDResult* res, float* msOut)
{
    uint8_t hdr[32]; memset(hdr, 0, sizeof(hdr));
    size_t const hn = f.srcLen < sizeof(hdr) ? f.srcLen : sizeof(hdr);
    const uint8_t* const hostFrame = c->bfHostSrc ? c->bfHostSrc + f.srcOff : nullptr;
    if (hostFrame) memcpy(hdr, hostFrame, hn);
    else if (hipMemcpyAsync(hdr, srcDev + f.srcOff, hn, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
    zhip::BfHeader const H = zhip::bf_parse_header(hdr, f.srcLen);
    if (!H.ok) return false;
    // the room the content may take: its stated size, or — a frame that does not state it — the destination slot (then the compressed size has to look large too)
    uint64_t const limit = H.known ? H.fcs : (uint64_t)f.dstCap;
    // (offsets of 2^31 and more would collide with the symbolic-history marker ZHIP_BF_SYM of the deferred offsets: content below 2 GiB keeps every real offset under it)
    if (limit < c->bigMin || limit > f.dstCap || limit >= 0x80000000ull || (!H.known && (uint64_t)f.srcLen * 64 < c->bigMin)) return false;
    const uint8_t* const src = srcDev + f.srcOff; uint8_t* const out = dstDev + f.dstOff;
    size_t const capBlocks = (size_t)(limit / 4096 + 4096);                 // a frame whose blocks regenerate less than 4 KB on average is left to k_decode
    if (!c->dBfInfo && hipMalloc((void**)&c->dBfInfo, sizeof(ZhipBfInfo)) != hipSuccess) return false;
    if (!bf_grow(c->dBfBlocks, c->bfBlocksCap, capBlocks)) return false;
    for (int i = 0; i < 2; i++) if (!c->bfEv[i] && hipEventCreate(&c->bfEv[i]) != hipSuccess) return false;
    ZhipBfInfo info; memset(&info, 0, sizeof(info));
    auto readInfo = [&]() { return hipMemcpyAsync(&info, c->dBfInfo, sizeof(info), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess; };
    (void)hipEventRecord(c->bfEv[0], s);
    if (hostFrame) {                                                   // the caller's bytes are in host memory: walk the block headers here (microseconds) and upload the table
        c->bfHostBlocks.resize(capBlocks);
        zhip::bf_walk_core(hostFrame, f.srcLen, H.hdrSize, H.blockMax, H.hasChecksum, c->bfHostBlocks.data(), (uint32_t)capBlocks, &info);
        if (info.status) return false;
        if (hipMemcpyAsync(c->dBfBlocks, c->bfHostBlocks.data(), (size_t)info.nBlocks * sizeof(ZhipBfBlock), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipMemcpyAsync(c->dBfInfo, &info, sizeof(info), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
    } else
    hipLaunchKernelGGL(zhip::k_bf_walk, dim3(1), dim3(64), 0, s, src, f.srcLen, H.hdrSize, H.blockMax, H.hasChecksum, c->dBfBlocks, (uint32_t)capBlocks, c->dBfInfo);
    hipLaunchKernelGGL(zhip::k_bf_prep, dim3((unsigned)((capBlocks + 255) / 256)), dim3(256), 0, s, src, H.blockMax, c->dBfBlocks, c->dBfInfo);
    hipLaunchKernelGGL(zhip::k_bf_deps, dim3(1), dim3(64), 0, s, c->dBfBlocks, c->dBfInfo);
    if (hipGetLastError() != hipSuccess || !readInfo() || info.status) return false;
    uint32_t const nB = info.nBlocks;
    if (nB == 0) return false;
    if (info.totalRecs > limit + nB) return false;                     // more sequences than bytes: not a frame worth 16 bytes per record
    if (!bf_grow(c->dBfLit, c->bfLitCap, (size_t)info.totalLit + 64) || !bf_grow(c->dBfRecs, c->bfRecsCap, (size_t)info.totalRecs + 1)) return false;
    hipLaunchKernelGGL(zhip::k_bf_entropy, dim3(nB), dim3(ZHIP_BF_THREADS), sizeof(zhip::DecShared), s, src, H.blockMax, c->dBfBlocks, c->dBfInfo, c->dBfLit, c->dBfRecs, c->dDefTabs);
    hipLaunchKernelGGL(zhip::k_bf_scan, dim3(1), dim3(64), 0, s, c->dBfBlocks, c->dBfInfo, (uint32_t)limit);       // the map has one entry per byte of the content: a frame that regenerates more than `limit` never reaches k_bf_build
    uint64_t content = H.fcs;
    if (!H.known) {                                                    // the blocks' sizes say how much content there is
        if (hipGetLastError() != hipSuccess || !readInfo() || info.status) return false;
        content = info.totalOut;
    }
    if (!bf_grow(c->dBfMap, c->bfMapCap, (size_t)content + 8)) return false;
    hipLaunchKernelGGL(zhip::k_bf_build, dim3(nB), dim3(256), 0, s, src, c->dBfBlocks, c->dBfInfo, c->dBfLit, c->dBfRecs, out, c->dBfMap);
    if (hipGetLastError() != hipSuccess || !readInfo() || info.status || info.totalOut != content) return false;
    uint32_t const n = (uint32_t)content, grid = (n + 1023) / 1024;
    unsigned rounds = 0;
    for (; n && rounds < 64; rounds++) {
        if (hipMemsetAsync(&c->dBfInfo->changed, 0, 4, s) != hipSuccess) return false;
        hipLaunchKernelGGL(zhip::k_bf_jump, dim3(grid), dim3(256), 0, s, c->dBfMap, n, c->dBfInfo);
        uint32_t changed = 1;
        if (hipMemcpyAsync(&changed, &c->dBfInfo->changed, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
        if (!changed) break;
    }
    if (rounds == 64) return false;
    if (n) hipLaunchKernelGGL(zhip::k_bf_copy, dim3(grid), dim3(256), 0, s, c->dBfMap, out, n);
    (void)hipEventRecord(c->bfEv[1], s);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
    float ms = 0; if (hipEventElapsedTime(&ms, c->bfEv[0], c->bfEv[1]) == hipSuccess) *msOut += ms;
    res->status = 0; res->size = n; res->hasChecksum = H.hasChecksum; res->checksum = info.checksum;
    c->bfLast[0]++; c->bfLast[2] += rounds; c->bfLast[3] += nB;
    // the arenas of a very large frame (the map alone is 4 bytes per content byte) do not stay with the context
    if (c->bfMapCap > ((size_t)1 << 30) + 4096) {              // content above 1 GiB: more than 4 GiB of map
        (void)hipFree(c->dBfMap); c->dBfMap = nullptr; c->bfMapCap = 0;
        (void)hipFree(c->dBfRecs); c->dBfRecs = nullptr; c->bfRecsCap = 0;
        (void)hipFree(c->dBfLit); c->dBfLit = nullptr; c->bfLitCap = 0;
    }
    return true;
}

// frames already described in c->hFrames[0..n); returns total decoded bytes or the first frame error
static size_t decode_locked(zhip_dctx* c, const zhip_ddict* dd, uint8_t* dstDev, const uint8_t* srcDev, size_t n, unsigned* statusOut,
                            unsigned long long* sizesOut, hipStream_t s)
{
    if (dd && dd->device != c->device) { snprintf(c->err, sizeof(c->err), "dictionary belongs to another device"); return DERR(1); }
    if (n == 0) return 0;
    if (n > 0xFFFFFFFFull) return DERR(72);
    ZhipDDictDev dv; memset(&dv, 0, sizeof(dv));
    if (dd) dv = dd->dev;
    // large frames first, one at a time over the whole GPU (zhip_decode_big.h); the others — and every frame that path declines — as a batch
    std::vector<ZhipDFrame> all; std::vector<ZhipDResult> bigRes; std::vector<size_t> rest;
    float bigMs = 0;
    c->bfLast[0] = c->bfLast[1] = c->bfLast[2] = c->bfLast[3] = 0;
    if (!dd && c->bigMin) {
        for (size_t i = 0; i < n; i++) {
            if (c->hFrames[i].dstCap < c->bigMin) continue;
            ZhipDResult r; memset(&r, 0, sizeof(r));
            if (!bigframe_decode(c, c->hFrames[i], dstDev, srcDev, s, &r, &bigMs)) { c->bfLast[1]++; continue; }
            if (all.empty()) { all.assign(c->hFrames, c->hFrames + n); bigRes.resize(n); for (size_t k = 0; k < n; k++) bigRes[k].status = 0xFFFFFFFFu; }
            bigRes[i] = r;
        }
    }
    size_t m = n;
    if (!all.empty()) { m = 0; for (size_t i = 0; i < n; i++) if (bigRes[i].status == 0xFFFFFFFFu) { rest.push_back(i); c->hFrames[m++] = all[i]; } }
    if (m) {
        DCHK(c, hipMemcpyAsync(c->dFrames, c->hFrames, m * sizeof(ZhipDFrame), hipMemcpyHostToDevice, s));
        DCHK(c, hipMemsetAsync(c->dCounter, 0, 4, s));
        uint32_t const grid = m < c->grid ? (uint32_t)m : c->grid;
        DCHK(c, hipEventRecord(c->ev[0], s));
        hipLaunchKernelGGL(zhip::k_decode, dim3(grid), dim3(ZHIP_DEC_THREADS), sizeof(zhip::DecShared), s,
                           srcDev, c->dFrames, (uint32_t)m, dstDev, c->dLit, c->dRecs, c->dCounter, dv, c->dDefTabs, c->dResults);
        DCHK(c, hipGetLastError());
        DCHK(c, hipEventRecord(c->ev[1], s));
        DCHK(c, hipMemcpyAsync(c->hResults, c->dResults, m * sizeof(ZhipDResult), hipMemcpyDeviceToHost, s));
        DCHK(c, hipStreamSynchronize(s));
    }
    if (!all.empty()) {                                         // back to the caller's order
        for (size_t k = m; k-- > 0; ) bigRes[rest[k]] = c->hResults[k];
        for (size_t i = 0; i < n; i++) { c->hFrames[i] = all[i]; c->hResults[i] = bigRes[i]; }
        if (n) DCHK(c, hipMemcpyAsync(c->dResults, c->hResults, n * sizeof(ZhipDResult), hipMemcpyHostToDevice, s));      // k_dec_verify works on the device copy
    }
    bool anyCheck = false;
    for (size_t i = 0; i < n; i++) if (c->hResults[i].hasChecksum) { anyCheck = true; break; }
    c->timing[1] = 0;
    if (anyCheck) {                                             // content checksums: XXH64 of every decoded frame, compared on the device
        for (size_t i = 0; i < n; i++) { ZhipUnit u; memset(&u, 0, sizeof(u)); u.srcOff = c->hFrames[i].dstOff; u.srcLen = c->hResults[i].hasChecksum ? c->hResults[i].size : 0; c->hUnits[i] = u; }
        DCHK(c, hipMemcpyAsync(c->dUnits, c->hUnits, n * sizeof(ZhipUnit), hipMemcpyHostToDevice, s));
        DCHK(c, hipEventRecord(c->ev[2], s));
        bool large = false;                                     // XXH64 is a serial chain per frame: a large one gets a whole wavefront (k_xxh64_wave), small ones share one
        for (size_t i = 0; i < n; i++) if (c->hUnits[i].srcLen >= (1u << 20)) { large = true; break; }
        if (large) hipLaunchKernelGGL(zhip::k_xxh64_wave, dim3((uint32_t)n), dim3(64), ZHIP_XXH_WAVE_LDS, s, (const uint8_t*)dstDev, c->dUnits, (uint32_t)n, c->dChecks);
        else
        hipLaunchKernelGGL(zhip::k_xxh64, dim3((uint32_t)((n + 15) / 16)), dim3(64), 0, s, (const uint8_t*)dstDev, c->dUnits, (uint32_t)n, c->dChecks);
        hipLaunchKernelGGL(zhip::k_dec_verify, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, c->dResults, c->dChecks, (uint32_t)n);
        DCHK(c, hipGetLastError());
        DCHK(c, hipEventRecord(c->ev[3], s));
        DCHK(c, hipMemcpyAsync(c->hResults, c->dResults, n * sizeof(ZhipDResult), hipMemcpyDeviceToHost, s));
        DCHK(c, hipStreamSynchronize(s));
        float ms = 0; if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) c->timing[1] = ms;
    }
    {   float ms = 0; if (m && hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->timing[0] = ms; else c->timing[0] = 0; c->timing[0] += bigMs; }
    uint64_t total = 0; size_t firstErr = 0;
    for (size_t i = 0; i < n; i++) {
        if (statusOut) statusOut[i] = c->hResults[i].status;
        if (sizesOut) sizesOut[i] = c->hResults[i].size;
        if (c->hResults[i].status && !firstErr) { firstErr = DERR(c->hResults[i].status); snprintf(c->err, sizeof(c->err), "frame %zu: zstd error %u", i, c->hResults[i].status); }
        total += c->hResults[i].size;
    }
    return firstErr ? firstErr : (size_t)total;
}

extern "C" {

size_t zhip_decompress_frames_device(zhip_dctx* c, const zhip_ddict* dd, void* dstDev, const unsigned long long* dstOffsets,
                                     const unsigned long long* dstCapacities, const void* srcDev, const unsigned long long* srcOffsets,
                                     const unsigned long long* srcSizes, size_t nFrames, unsigned* statusOut, unsigned long long* sizesOut, void* stream)
{
    std::lock_guard<std::mutex> lk(c->mu);
    DCHK(c, hipSetDevice(c->device));
    size_t const e = ensure_frames(c, nFrames);
    if (e) return e;
    for (size_t i = 0; i < nFrames; i++) {
        if (srcSizes[i] > 0xFFFFFFFFull) return DERR(72);
        ZhipDFrame f; f.srcOff = srcOffsets[i]; f.dstOff = dstOffsets[i]; f.srcLen = (uint32_t)srcSizes[i];
        f.dstCap = dstCapacities[i] > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dstCapacities[i];
        c->hFrames[i] = f;
    }
    return decode_locked(c, dd, (uint8_t*)dstDev, (const uint8_t*)srcDev, nFrames, statusOut, sizesOut, stream ? (hipStream_t)stream : c->stream);
}

// = ZSTD_seekable_decompress (contrib/seekable_format/zstd_seekable.h, zstdseek_decompress.c:ZSTD_seekable_decompress): `len`
// bytes of the decompressed data starting at `offset`, from a seekable file held in a HOST buffer.  The seek table (the
// skippable frame at the end, contrib/seekable_format/zstd_seekable_compression_format.md) names every frame's compressed and
// decompressed size; only the frames that overlap the request are staged and decoded — as one batch, one workgroup each —
// and their stored checksums (XXH64 low words, when the table has them) are verified against the decoded content.
size_t zhip_seekable_read(zhip_dctx* c, void* dst, size_t len, const void* srcv, size_t srcSize, unsigned long long offset)
{
    std::lock_guard<std::mutex> lk(c->mu);
    DCHK(c, hipSetDevice(c->device));
    const uint8_t* const src = (const uint8_t*)srcv;
    auto rd32 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    if (srcSize < 17 || rd32(src + srcSize - 4) != 0x8F92EAB1u) return DERR(10);                    // no seek table footer: prefix_unknown
    uint32_t const nFrames = rd32(src + srcSize - 9); uint8_t const desc = src[srcSize - 5];
    if (desc & 0x7C) return DERR(20);                                                                 // reserved bits (zstdseek_decompress.c)
    size_t const entry = (desc & 0x80) ? 12 : 8, tableSize = 8 + (size_t)nFrames * entry + 9;
    if (tableSize > srcSize) return DERR(20);
    const uint8_t* const tab = src + srcSize - tableSize;
    if (rd32(tab) != 0x184D2A5Eu || rd32(tab + 4) != (uint32_t)(tableSize - 8)) return DERR(20);
    if (len == 0) return 0;
    // frames overlapping [offset, offset + len)
    uint64_t cpos = 0, dpos = 0; size_t first = nFrames, last = 0; uint64_t firstC = 0, firstD = 0;
    for (size_t i = 0; i < nFrames; i++) {
        uint32_t const cs = rd32(tab + 8 + i * entry), ds = rd32(tab + 8 + i * entry + 4);
        if (dpos + ds > offset && dpos < offset + len && ds) { if (first == nFrames) { first = i; firstC = cpos; firstD = dpos; } last = i + 1; }
        cpos += cs; dpos += ds;
    }
    if (cpos + tableSize > srcSize) return DERR(20);
    if (offset + len > dpos || first == nFrames) return DERR(72);                                     // request beyond the end: srcSize_wrong
    size_t const n = last - first;
    size_t const e = ensure_frames(c, n);
    if (e) return e;
    uint64_t so = 0, d0 = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t const cs = rd32(tab + 8 + (first + i) * entry), ds = rd32(tab + 8 + (first + i) * entry + 4);
        ZhipDFrame f; f.srcOff = so; f.dstOff = d0; f.srcLen = cs; f.dstCap = ds;
        c->hFrames[i] = f; so += cs; d0 += ds;
    }
    if (c->srcStageCap < so + 64) { (void)hipFree(c->dSrcStage); c->dSrcStage = nullptr; c->srcStageCap = 0; DCHK(c, hipMalloc((void**)&c->dSrcStage, so + 64)); c->srcStageCap = so + 64; }
    if (c->dstStageCap < d0 + 64) { (void)hipFree(c->dDstStage); c->dDstStage = nullptr; c->dstStageCap = 0; DCHK(c, hipMalloc((void**)&c->dDstStage, d0 + 64)); c->dstStageCap = d0 + 64; }
    DCHK(c, hipMemcpyAsync(c->dSrcStage, src + firstC, so, hipMemcpyHostToDevice, c->stream));
    size_t const r = decode_locked(c, nullptr, c->dDstStage, c->dSrcStage, n, nullptr, nullptr, c->stream);
    if (zhip_isError(r)) return r;
    for (size_t i = 0; i < n; i++) if (c->hResults[i].size != c->hFrames[i].dstCap) return DERR(20);      // the table and the frame disagree
    if (desc & 0x80) {                                                                                // the table's own checksums
        for (size_t i = 0; i < n; i++) { ZhipUnit u; memset(&u, 0, sizeof(u)); u.srcOff = c->hFrames[i].dstOff; u.srcLen = c->hResults[i].size; c->hUnits[i] = u; }
        DCHK(c, hipMemcpyAsync(c->dUnits, c->hUnits, n * sizeof(ZhipUnit), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(zhip::k_xxh64, dim3((uint32_t)((n + 15) / 16)), dim3(64), 0, c->stream, (const uint8_t*)c->dDstStage, c->dUnits, (uint32_t)n, c->dChecks);
        DCHK(c, hipGetLastError());
        std::vector<uint32_t> ck(n);
        DCHK(c, hipMemcpyAsync(ck.data(), c->dChecks, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        DCHK(c, hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < n; i++) if (ck[i] != rd32(tab + 8 + (first + i) * entry + 8)) return DERR(22);
    }
    DCHK(c, hipMemcpy(dst, c->dDstStage + (offset - firstD), len, hipMemcpyDeviceToHost));
    return len;
}

// = ZSTD_decompress / ZSTD_decompress_usingDDict (lib/zstd.h:168, :1013) for host buffers: every frame of src, contents back to back in dst
size_t zhip_decompress(zhip_dctx* c, const zhip_ddict* dd, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    std::lock_guard<std::mutex> lk(c->mu);
    DCHK(c, hipSetDevice(c->device));
    size_t const n = zhip_find_frames(src, srcSize, nullptr, nullptr, nullptr, nullptr, 0);
    if (zhip_isError(n)) return n;
    if (n == 0) return 0;
    std::vector<unsigned long long> so(n), ss(n), cs(n), cb(n);
    (void)zhip_find_frames(src, srcSize, so.data(), ss.data(), cs.data(), cb.data(), n);
    size_t const e = ensure_frames(c, n);
    if (e) return e;
    // destination slots: the stated content size, or the frame's bound when the header does not hold it — but never more than
    // the caller's buffer could take (a hostile 1 MB input of tiny RLE blocks "bounds" to tens of GB: the reference never
    // allocates beyond dst, neither does this), and frames without a stated size are decoded in groups of bounded staging
    bool exact = true;
    for (size_t i = 0; i < n; i++) if (cs[i] == ~0ull) exact = false;
    if (c->srcStageCap < srcSize + 64) { (void)hipFree(c->dSrcStage); c->dSrcStage = nullptr; c->srcStageCap = 0; DCHK(c, hipMalloc((void**)&c->dSrcStage, srcSize + 64)); c->srcStageCap = srcSize + 64; }
    DCHK(c, hipMemcpyAsync(c->dSrcStage, src, srcSize, hipMemcpyHostToDevice, c->stream));
    uint64_t const budget = exact ? ~0ull : (2 * (uint64_t)dstCapacity > ((uint64_t)1 << 20) ? 2 * (uint64_t)dstCapacity : ((uint64_t)1 << 20));
    size_t pos = 0, i0 = 0;
    while (i0 < n) {
        uint64_t total = 0; size_t g = 0;
        while (i0 + g < n) {
            size_t const i = i0 + g;
            uint64_t room = cs[i] != ~0ull ? cs[i] : cb[i];
            if (cs[i] == ~0ull && room > dstCapacity) room = dstCapacity;          // a frame that needs more gets dstSize_tooSmall from the kernel
            if (room > 0xFFFFFFFFull || ss[i] > 0xFFFFFFFFull) return DERR(14);
            if (g && total + room > budget) break;
            ZhipDFrame f; f.srcOff = so[i]; f.dstOff = total; f.srcLen = (uint32_t)ss[i]; f.dstCap = (uint32_t)room;
            c->hFrames[g] = f; total += room; g++;
        }
        if (exact && total > dstCapacity) return DERR(70);
        if (c->dstStageCap < total + 64) { (void)hipFree(c->dDstStage); c->dDstStage = nullptr; c->dstStageCap = 0; DCHK(c, hipMalloc((void**)&c->dDstStage, total + 64)); c->dstStageCap = total + 64; }
        c->bfHostSrc = (const uint8_t*)src;                         // the staged copy holds the same bytes at the same offsets
        size_t const r = decode_locked(c, dd, c->dDstStage, c->dSrcStage, g, nullptr, nullptr, c->stream);
        c->bfHostSrc = nullptr;
        if (zhip_isError(r)) return r;
        if (r > dstCapacity - pos) return DERR(70);
        if (exact) { if (r) DCHK(c, hipMemcpy((uint8_t*)dst + pos, c->dDstStage, r, hipMemcpyDeviceToHost)); pos += r; }
        else {                                                  // bound-sized slots: pack
            for (size_t k = 0; k < g; k++) {
                size_t const sz = c->hResults[k].size;
                if (sz) DCHK(c, hipMemcpy((uint8_t*)dst + pos, c->dDstStage + c->hFrames[k].dstOff, sz, hipMemcpyDeviceToHost));
                pos += sz;
            }
        }
        i0 += g;
    }
    return pos;
}

}  // extern "C"
