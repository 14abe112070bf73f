// zhip_multi.h — in-process multi-device compression of HOST buffers (north_star: "independent blocks/frames shard
// across the GPUs of one node on separate HIP streams with a host-side gather — no RCCL collectives").  Host C++ only.
//
// The source is cut into chunks of `chunkUnits` units; chunk k belongs to lane (k mod nLanes), nLanes = ZHIP_MULTI_LANES (default 4)
// per device, each lane a host thread with its own zhip_ctx + HIP stream + pinned staging, so that on every device some lanes'
// host memcpy and H2D / D2H copies overlap another lane's kernels (multi-buffering; measured on one MI355X with a 1 GiB source:
// 2 lanes 12.8 GB/s, 4 lanes 26.9 GB/s — the pageable-to-pinned memcpy of one host thread is the slowest stage), and devices
// run independently.  No exchange step exists:
// units are independent.  Results are variable-length, so a finished chunk publishes its size, and is copied to its final
// place as soon as every earlier chunk (in source order) has published: the destination offset of chunk k is the exclusive
// prefix sum of the sizes before it — the ordered host gather.
#pragma once
#include <condition_variable>
#include <thread>
#include <vector>

struct zhip_multi_lane {
    int device = 0;
    zhip_ctx* ctx = nullptr;
    uint8_t *pinIn = nullptr, *pinOut = nullptr;       // hipHostMalloc
    uint32_t* pinSizes = nullptr;
    uint8_t *dIn = nullptr, *dOut = nullptr; uint32_t* dSizes = nullptr;
    size_t inCap = 0, outCap = 0;
};
struct zhip_multi_s {
    std::vector<zhip_multi_lane> lanes;
    size_t chunkUnits = 0;
    int checksum = 0;
    std::mutex mu;                                     // one call at a time
    char err[256] = {0};
    double lastSeconds = 0;
};

static void multi_free(zhip_multi_s* m)
{
    for (auto& L : m->lanes) {
        (void)hipSetDevice(L.device);
        if (L.ctx) zhip_destroy(L.ctx);
        (void)hipHostFree(L.pinIn); (void)hipHostFree(L.pinOut); (void)hipHostFree(L.pinSizes);
        (void)hipFree(L.dIn); (void)hipFree(L.dOut); (void)hipFree(L.dSizes);
    }
    delete m;
}

extern "C" {

zhip_multi* zhip_multi_create(const int* devices, int nDevices, size_t chunkUnits)
{
    if (!devices || nDevices <= 0) return nullptr;
    if (chunkUnits == 0) chunkUnits = 512;                                 // 64 MB of source per chunk
    size_t lanesPer = 4;
    if (const char* e = getenv("ZHIP_MULTI_LANES")) { long const v = atol(e); if (v >= 1 && v <= 16) lanesPer = (size_t)v; }
    zhip_multi_s* m = new zhip_multi_s();
    m->chunkUnits = chunkUnits;
    size_t const inCap = chunkUnits * (size_t)ZHIP_UNIT_MAX + 64, outCap = zhip_compressBound(chunkUnits * (size_t)ZHIP_UNIT_MAX, ZHIP_UNIT_MAX) + 64;
    m->lanes.resize(lanesPer * (size_t)nDevices);
    for (size_t i = 0; i < m->lanes.size(); i++) {
        zhip_multi_lane& L = m->lanes[i];
        L.device = devices[i / lanesPer]; L.inCap = inCap; L.outCap = outCap;
        bool ok = hipSetDevice(L.device) == hipSuccess;
        ok = ok && (L.ctx = zhip_create(L.device, chunkUnits)) != nullptr;
        ok = ok && hipHostMalloc((void**)&L.pinIn, inCap, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&L.pinOut, outCap, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipHostMalloc((void**)&L.pinSizes, chunkUnits * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
        ok = ok && hipMalloc((void**)&L.dIn, inCap) == hipSuccess && hipMalloc((void**)&L.dOut, outCap) == hipSuccess;
        ok = ok && hipMalloc((void**)&L.dSizes, chunkUnits * sizeof(uint32_t)) == hipSuccess;
        if (!ok) { multi_free(m); return nullptr; }
    }
    return m;
}

void zhip_multi_destroy(zhip_multi* m) { if (m) multi_free(m); }
int zhip_multi_set_frame_checksum(zhip_multi* m, int enable) { std::lock_guard<std::mutex> lk(m->mu); m->checksum = enable ? 1 : 0; return 0; }
const char* zhip_multi_last_error(const zhip_multi* m) { return m->err; }
double zhip_multi_last_seconds(const zhip_multi* m) { return m->lastSeconds; }

size_t zhip_compress_multi(zhip_multi* m, void* dstv, size_t dstCapacity, const void* srcv, size_t srcSize,
                           int level, const unsigned cparams[7], size_t unitSize, size_t* unitSizes)
{
    std::lock_guard<std::mutex> lk(m->mu);
    if (unitSize == 0 || unitSize > ZHIP_UNIT_MAX) return ZERR(ZE_parameter_outOfBound);
    if (cparams && !zhip::host_check_overrides(cparams)) return ZERR(ZE_parameter_outOfBound);
    auto const t0 = std::chrono::steady_clock::now();
    uint8_t* const dst = (uint8_t*)dstv; const uint8_t* const src = (const uint8_t*)srcv;
    size_t const chunkBytes = m->chunkUnits * unitSize;
    size_t const nChunks = srcSize ? (srcSize + chunkBytes - 1) / chunkBytes : 1;
    size_t const nLanes = m->lanes.size();
    // ordered gather state
    std::vector<size_t> size(nChunks, 0), off(nChunks + 1, 0);
    std::vector<char> known(nChunks, 0);
    size_t placed = 0;                                   // chunks [0, placed) have their offsets
    size_t firstErr = 0;
    std::mutex gm; std::condition_variable gcv;
    m->err[0] = 0;

    auto lane_fn = [&](size_t li) {
        zhip_multi_lane& L = m->lanes[li];
        if (hipSetDevice(L.device) != hipSuccess) { std::lock_guard<std::mutex> g(gm); if (!firstErr) firstErr = ZERR(ZE_GENERIC); gcv.notify_all(); return; }
        zhip_set_frame_checksum(L.ctx, m->checksum);
        for (size_t k = li; k < nChunks; k += nLanes) {
            {   std::lock_guard<std::mutex> g(gm); if (firstErr) return; }
            size_t const b0 = k * chunkBytes, len = srcSize - b0 < chunkBytes ? srcSize - b0 : chunkBytes;
            if (len) memcpy(L.pinIn, src + b0, len);
            size_t r = 0;
            hipStream_t const s = L.ctx->stream;
            if (len && hipMemcpyAsync(L.dIn, L.pinIn, len, hipMemcpyHostToDevice, s) != hipSuccess) r = ZERR(ZE_GENERIC);
            if (!r) r = zhip_compress_params_device(L.ctx, L.dOut, L.outCap, L.dIn, len, level, cparams, unitSize, L.dSizes, (void*)s);
            size_t const nu = len ? (len + unitSize - 1) / unitSize : 1;
            if (!zhip_isError(r)) {
                bool ok = hipMemcpyAsync(L.pinOut, L.dOut, r, hipMemcpyDeviceToHost, s) == hipSuccess;
                ok = ok && hipMemcpyAsync(L.pinSizes, L.dSizes, nu * sizeof(uint32_t), hipMemcpyDeviceToHost, s) == hipSuccess;
                ok = ok && hipStreamSynchronize(s) == hipSuccess;
                if (!ok) r = ZERR(ZE_GENERIC);
            }
            size_t myOff = 0;
            {   std::unique_lock<std::mutex> g(gm);
                if (zhip_isError(r)) { if (!firstErr) { firstErr = r; snprintf(m->err, sizeof(m->err), "chunk %zu on device %d: %s", k, L.device, zhip_last_error(L.ctx)); } gcv.notify_all(); return; }
                size[k] = r; known[k] = 1;
                while (placed < nChunks && known[placed]) { off[placed + 1] = off[placed] + size[placed]; placed++; }
                gcv.notify_all();
                gcv.wait(g, [&] { return placed > k || firstErr; });
                if (firstErr) return;
                myOff = off[k];
                if (myOff + r > dstCapacity) { firstErr = ZERR(ZE_dstSize_tooSmall); gcv.notify_all(); return; }
            }
            memcpy(dst + myOff, L.pinOut, r);
            if (unitSizes) { size_t const u0 = k * m->chunkUnits; for (size_t i = 0; i < nu; i++) unitSizes[u0 + i] = L.pinSizes[i]; }
        }
    };
    std::vector<std::thread> th;
    size_t const use = nChunks < nLanes ? nChunks : nLanes;
    for (size_t li = 1; li < use; li++) th.emplace_back(lane_fn, li);
    lane_fn(0);
    for (auto& t : th) t.join();
    m->lastSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (firstErr) return firstErr;
    return off[nChunks];
}

}  // extern "C"
