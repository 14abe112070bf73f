// zhip_multi.h — in-process multi-device compression of HOST buffers (north_star: "independent blocks/frames shard
// across the GPUs of one node on separate HIP streams with a host-side gather — no RCCL collectives").  Host C++ only.
//
// The source is cut into chunks of `chunkUnits` units; chunk k belongs to lane (k mod nLanes), nLanes = ZHIP_MULTI_LANES (default 2)
// per device.  A lane is one zhip_ctx + HIP stream and three host threads with two pinned slots between them in each direction: a feeder
// copies the next chunk from the caller's pageable buffer into staging while the device thread runs H2D -> kernels -> D2H of the current
// one and a gatherer copies finished frames to their final place (round 4: one thread per lane did all of it in turn, 22 GB/s; now
// 29 GB/s on one MI355X with a 1 GiB source — the per-lane chain H2D + kernels + D2H of a 128 MB chunk, 7.3 ms, is what is left), and devices
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
    uint8_t *pinIn[2] = {nullptr, nullptr}, *pinOut[2] = {nullptr, nullptr};       // hipHostMalloc; two slots each: the copy into / out of staging of one chunk runs beside the device work of another
    uint32_t* pinSizes[2] = {nullptr, nullptr};
    uint8_t *dIn = nullptr, *dOut = nullptr; uint32_t* dSizes = nullptr;
    uint8_t* dIn2[2] = {nullptr, nullptr};                     // zhip_compress_multi: two device input buffers (dIn2[0] == dIn), so that the H2D of chunk i+1 runs under the kernels of chunk i
    hipStream_t copyStream = nullptr;                          // the lane's H2D copies (the kernels and the D2H are on ctx->stream)
    hipEvent_t evIn[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // per input slot: before / after its H2D on copyStream
    size_t inCap = 0, outCap = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // stage marks of the chunk in flight: (unused), start of the kernels, after the kernels, after D2H
};
struct zhip_multi_s {
    std::vector<zhip_multi_lane> lanes;
    size_t chunkUnits = 0;
    int checksum = 0;
    int rowMode = -1;                                   // zhip_set_row_matcher mode as last set (-1 = every context's own default)
    std::mutex mu;                                     // one call at a time
    char err[256] = {0};
    double lastSeconds = 0;
    double stages[7] = {0, 0, 0, 0, 0, 0, 0};          // zhip_multi_last_stages
    zhip_ctx* wide = nullptr; size_t wideUnits = 0;    // job-pool frames that go to ONE context (checksum, jobs larger than a staging buffer): sized on demand
};

static void multi_free(zhip_multi_s* m)
{
    if (m->wide) zhip_destroy(m->wide);
    for (auto& L : m->lanes) {
        (void)hipSetDevice(L.device);
        if (L.ctx) zhip_destroy(L.ctx);
        for (auto& e : L.ev) if (e) (void)hipEventDestroy(e);
        for (auto& ee : L.evIn) for (auto& e : ee) if (e) (void)hipEventDestroy(e);
        if (L.copyStream) (void)hipStreamDestroy(L.copyStream);
        (void)hipFree(L.dIn2[1]);
        for (int b = 0; b < 2; b++) { (void)hipHostFree(L.pinIn[b]); (void)hipHostFree(L.pinOut[b]); (void)hipHostFree(L.pinSizes[b]); }
        (void)hipFree(L.dIn); (void)hipFree(L.dOut); (void)hipFree(L.dSizes);
    }
    delete m;
}

extern "C" {

zhip_multi* zhip_multi_create(const int* devices, int nDevices, size_t chunkUnits)
{
    if (!devices || nDevices <= 0) return nullptr;
    {   // every ordinal must exist BEFORE anything is touched: a failed hipSetDevice leaves a sticky "invalid device ordinal" behind that
        // the next, unrelated, launch check would report
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        for (int i = 0; i < nDevices; i++) if (devices[i] < 0 || devices[i] >= count) return nullptr;
    }
    // 128 MB of source per chunk on two lanes per device (profiles/r05_e2e_stages.log, 1 GiB, level 1, one MI355X): a chunk's kernels have a
    // floor of about 3.3 ms (one unit is a serial walk), so small chunks waste the device (512 units: 3.3 ms, 1024: 3.75 ms) and more than two
    // chunks' kernels at once only queue behind each other (per-chunk kernel time 3.3 / 6.4 / 11.2 ms at 2 / 4 / 8 lanes)
    if (chunkUnits == 0) chunkUnits = 1024;
    size_t lanesPer = 2;
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
#ifdef ZHIP_MULTI_SLOT_SHARE
        if (ok) L.ctx->slotShare = (int)lanesPer;
#endif
        for (int b = 0; b < 2; b++) {
            ok = ok && hipHostMalloc((void**)&L.pinIn[b], inCap, hipHostMallocDefault) == hipSuccess;
            ok = ok && hipHostMalloc((void**)&L.pinOut[b], outCap, hipHostMallocDefault) == hipSuccess;
            ok = ok && hipHostMalloc((void**)&L.pinSizes[b], chunkUnits * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
        }
        ok = ok && hipMalloc((void**)&L.dIn, inCap) == hipSuccess && hipMalloc((void**)&L.dOut, outCap) == hipSuccess;
        ok = ok && hipMalloc((void**)&L.dSizes, chunkUnits * sizeof(uint32_t)) == hipSuccess;
        for (auto& e : L.ev) ok = ok && hipEventCreate(&e) == hipSuccess;
        for (auto& ee : L.evIn) for (auto& e : ee) ok = ok && hipEventCreate(&e) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&L.copyStream, hipStreamNonBlocking) == hipSuccess;
        L.dIn2[0] = L.dIn;
        ok = ok && hipMalloc((void**)&L.dIn2[1], inCap) == hipSuccess;
        if (!ok) { multi_free(m); (void)hipGetLastError(); return nullptr; }
    }
    return m;
}

void zhip_multi_destroy(zhip_multi* m) { if (m) multi_free(m); }
int zhip_multi_set_frame_checksum(zhip_multi* m, int enable) { std::lock_guard<std::mutex> lk(m->mu); m->checksum = enable ? 1 : 0; return 0; }
// ZSTD_c_useRowMatchFinder for every lane context (and the wide one, when it exists or is made later)
int zhip_multi_set_row_matcher(zhip_multi* m, int mode)
{
    std::lock_guard<std::mutex> lk(m->mu);
    if (mode < -1 || mode > 2) return 1;
    m->rowMode = mode;
    for (auto& L : m->lanes) if (L.ctx) zhip_set_row_matcher(L.ctx, mode);
    if (m->wide) zhip_set_row_matcher(m->wide, mode);
    return 0;
}
const char* zhip_multi_last_error(const zhip_multi* m) { return m->err; }
double zhip_multi_last_seconds(const zhip_multi* m) { return m->lastSeconds; }
void zhip_multi_last_stages(const zhip_multi* m, double out[7]) { for (int i = 0; i < 7; i++) out[i] = m->stages[i]; }

#ifndef ZHIP_MULTI_COPY_THREADS
#define ZHIP_MULTI_COPY_THREADS 4
#endif
size_t zhip_compress_multi(zhip_multi* m, void* dstv, size_t dstCapacity, const void* srcv, size_t srcSize,
                           int level, const unsigned cparams[7], size_t unitSize, size_t* unitSizes)
{
    std::lock_guard<std::mutex> lk(m->mu);
    if (unitSize == 0 || unitSize > ZHIP_UNIT_MAX) return ZERR(ZE_parameter_outOfBound);
    if (cparams && !zhip::host_check_overrides(cparams)) return ZERR(ZE_parameter_outOfBound);
    auto const t0 = std::chrono::steady_clock::now();
    uint8_t* const dst = (uint8_t*)dstv; const uint8_t* const src = (const uint8_t*)srcv;
    size_t const nLanes = m->lanes.size();
    // The chunk plan.  Full chunks of chunkUnits units, with a RAMP at both ends when the source is long enough: the first chunk of every lane
    // is a quarter chunk (the device starts after 32 MB have been copied into staging instead of 128), and so is the last one (what runs after the
    // last byte has crossed PCIe — kernels, D2H, copy-out of one chunk — is a quarter as long).  cu0[k] = first unit of chunk k.
    size_t const nUnitsAll = srcSize ? (srcSize + unitSize - 1) / unitSize : 1;
    std::vector<size_t> cu0;
    {   size_t const full = m->chunkUnits, small = full >= 8 ? full / 4 : full;
        bool const ramp = small < full && nUnitsAll >= 2 * nLanes * full;
        size_t u = 0, k = 0;
        size_t const tailStart = ramp ? nUnitsAll - nLanes * small : nUnitsAll;
        while (u < nUnitsAll) {
            cu0.push_back(u);
            size_t take = (ramp && (k < nLanes || u >= tailStart)) ? small : full;
            if (ramp && u < tailStart && u + take > tailStart) take = tailStart - u;     // the last full-size chunk ends where the tail ramp starts
            u += take; k++;
        }
        cu0.push_back(nUnitsAll);
    }
    size_t const nChunks = cu0.size() - 1;
    auto chunk_b0 = [&](size_t k) { return cu0[k] * unitSize; };
    auto chunk_len = [&](size_t k) { size_t const a = cu0[k] * unitSize, b = cu0[k + 1] * unitSize; return (b < srcSize ? b : srcSize) - (a < srcSize ? a : srcSize); };
    // ordered gather state
    std::vector<size_t> size(nChunks, 0), off(nChunks + 1, 0);
    std::vector<char> known(nChunks, 0);
    size_t placed = 0;                                   // chunks [0, placed) have their offsets
    size_t firstErr = 0;
    std::mutex gm; std::condition_variable gcv;
    m->err[0] = 0;
    for (auto& v : m->stages) v = 0;

    // A lane is three host threads around two HIP streams, two pinned slots between them in each direction:
    //   feeder  : copies chunk i+1 of the lane from the caller's (pageable) buffer into a free input slot and issues its H2D on the lane's copy
    //             stream (its own device buffer), while chunk i is in its kernels;
    //   device  : kernels -> D2H of one chunk at a time on the lane's stream, behind the event of the chunk's H2D;
    //   gatherer: waits until every earlier chunk (in source order) has published its size, then copies the frames to their final place.
    // (Round 4 ran the three in one thread per lane: 10.4 ms per 64 MB chunk of which 4 ms were the device's — profiles/r05_e2e_stages.log.)
    // a staging copy of more than a few MB is split over ZHIP_MULTI_COPY_THREADS host threads (one thread moves ~26 GB/s of pageable memory on this
    // box: 4.6 ms per 128 MB chunk, as long as the chunk's kernels)
    auto par_copy = [](uint8_t* d, const uint8_t* s_, size_t len) {
        size_t const parts = len >= ((size_t)8 << 20) ? ZHIP_MULTI_COPY_THREADS : 1;
        if (parts <= 1) { if (len) memcpy(d, s_, len); return; }
        size_t const per = ((len / parts) + 4095) & ~(size_t)4095;
        std::vector<std::thread> hs;
        try { for (size_t i = 1; i < parts; i++) { size_t const a = i * per; if (a < len) hs.emplace_back([=] { memcpy(d + a, s_ + a, (a + per < len ? a + per : len) - a); }); } }
        catch (...) { for (auto& h : hs) h.join(); memcpy(d, s_, len); return; }        // no thread to be had: one copy of everything (idempotent)
        memcpy(d, s_, per < len ? per : len);
        for (auto& h : hs) h.join();
    };
    auto lane_fn = [&](size_t li) {
        zhip_multi_lane& L = m->lanes[li];
        std::vector<size_t> mine;
        for (size_t k = li; k < nChunks; k += nLanes) mine.push_back(k);
        std::mutex lm; std::condition_variable lcv;
        int inState[2] = {0, 0};            // 0 free, 1 filled
        int outState[2] = {0, 0};           // 0 free, 1 holds a finished chunk
        size_t outBytes[2] = {0, 0};
        bool stop = false;
        // (the context's own message is only read by the lane's DEVICE thread, the one that may be writing it; the feeder and the gatherer name their stage)
        auto fail = [&](size_t code, size_t k, const char* stage = nullptr) {
            {   std::lock_guard<std::mutex> g(gm);
                if (!firstErr) { firstErr = code; snprintf(m->err, sizeof(m->err), "chunk %zu on device %d: %s", k, L.device, stage ? stage : zhip_last_error(L.ctx)); }
                gcv.notify_all(); }
            {   std::lock_guard<std::mutex> g(lm); stop = true; }
            lcv.notify_all();
        };
        auto failed = [&] { std::lock_guard<std::mutex> g(gm); return firstErr != 0; };
        double tIn = 0, tWait = 0, tOut = 0, tH2D = 0, tK = 0, tD2H = 0;
        auto feeder_fn = [&] {
            for (size_t i = 0; i < mine.size(); i++) {
                int const b = (int)(i & 1);
                {   std::unique_lock<std::mutex> g(lm); lcv.wait(g, [&] { return inState[b] == 0 || stop; }); if (stop) return; }
                size_t const k = mine[i], b0 = chunk_b0(k), len = chunk_len(k);
                auto const t0 = std::chrono::steady_clock::now();
                par_copy(L.pinIn[b], src + b0, len);
                tIn += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                bool ok = i != 0 || hipSetDevice(L.device) == hipSuccess;
                ok = ok && hipEventRecord(L.evIn[b][0], L.copyStream) == hipSuccess;
                if (ok && len) ok = hipMemcpyAsync(L.dIn2[b], L.pinIn[b], len, hipMemcpyHostToDevice, L.copyStream) == hipSuccess;
                ok = ok && hipEventRecord(L.evIn[b][1], L.copyStream) == hipSuccess;
                if (!ok) { fail(ZERR(ZE_GENERIC), k, "the host-to-device copy could not be issued"); return; }
                {   std::lock_guard<std::mutex> g(lm); inState[b] = 1; }
                lcv.notify_all();
            }
        };
        auto gatherer_fn = [&] {
            for (size_t i = 0; i < mine.size(); i++) {
                int const b = (int)(i & 1);
                size_t const k = mine[i];
                {   std::unique_lock<std::mutex> g(lm); lcv.wait(g, [&] { return outState[b] == 1 || stop; }); if (stop) return; }
                size_t const r = outBytes[b];
                size_t const len = chunk_len(k);
                size_t const nu = len ? (len + unitSize - 1) / unitSize : 1;
                auto const t0 = std::chrono::steady_clock::now();
                size_t myOff = 0;
                {   std::unique_lock<std::mutex> g(gm);
                    size[k] = r; known[k] = 1;
                    while (placed < nChunks && known[placed]) { off[placed + 1] = off[placed] + size[placed]; placed++; }
                    gcv.notify_all();
                    gcv.wait(g, [&] { return placed > k || firstErr; });
                    if (firstErr) { g.unlock(); { std::lock_guard<std::mutex> g2(lm); stop = true; } lcv.notify_all(); return; }
                    myOff = off[k];
                }
                if (myOff + r > dstCapacity) { fail(ZERR(ZE_dstSize_tooSmall), k, "the destination buffer is too small for the frames"); return; }
                auto const t1 = std::chrono::steady_clock::now();
                par_copy(dst + myOff, L.pinOut[b], r);
                if (unitSizes) { size_t const u0 = cu0[k]; for (size_t j = 0; j < nu; j++) unitSizes[u0 + j] = L.pinSizes[b][j]; }
                auto const t2 = std::chrono::steady_clock::now();
                tWait += std::chrono::duration<double>(t1 - t0).count(); tOut += std::chrono::duration<double>(t2 - t1).count();
                {   std::lock_guard<std::mutex> g(lm); outState[b] = 0; }
                lcv.notify_all();
            }
        };
        // a thread that cannot be had must not take the process down from inside a C call (std::system_error out of a std::thread constructor on a lane
        // thread is std::terminate): the call fails with memory_allocation instead and whatever started is joined (round-5 advisor finding)
        std::thread feeder, gatherer;
        try { feeder = std::thread(feeder_fn); gatherer = std::thread(gatherer_fn); }
        catch (...) {
            fail(ZERR(ZE_memory_allocation), mine.empty() ? 0 : mine[0], "no host thread for the lane's feeder / gatherer");
            if (feeder.joinable()) feeder.join();
            if (gatherer.joinable()) gatherer.join();
            return;
        }
        if (hipSetDevice(L.device) != hipSuccess) fail(ZERR(ZE_GENERIC), mine.empty() ? 0 : mine[0], "hipSetDevice failed");
        else {
            zhip_set_frame_checksum(L.ctx, m->checksum);
            for (size_t i = 0; i < mine.size(); i++) {
                int const b = (int)(i & 1);
                size_t const k = mine[i], len = chunk_len(k);
                if (failed()) { { std::lock_guard<std::mutex> g(lm); stop = true; } lcv.notify_all(); break; }
                {   std::unique_lock<std::mutex> g(lm); lcv.wait(g, [&] { return (inState[b] == 1 && outState[b] == 0) || stop; }); if (stop) break; }
                size_t r = 0;
                hipStream_t const s = L.ctx->stream;
                if (hipStreamWaitEvent(s, L.evIn[b][1], 0) != hipSuccess) r = ZERR(ZE_GENERIC);       // the chunk's H2D (issued by the feeder on the copy stream)
                (void)hipEventRecord(L.ev[1], s);
                if (!r) r = zhip_compress_params_device(L.ctx, L.dOut, L.outCap, L.dIn2[b], len, level, cparams, unitSize, L.dSizes, (void*)s);
                size_t const nu = len ? (len + unitSize - 1) / unitSize : 1;
                if (!zhip_isError(r)) {
                    // (the D2H stays on the kernels' stream: on its own stream, under the next chunk's kernels, the call got slower — 30.0 -> 24.6 GB/s,
                    // per-chunk kernel time 3.7 -> 6.2 ms, profiles/r05_e2e_stages.log)
                    bool ok = hipEventRecord(L.ev[2], s) == hipSuccess;
                    ok = ok && hipMemcpyAsync(L.pinOut[b], L.dOut, r, hipMemcpyDeviceToHost, s) == hipSuccess;
                    ok = ok && hipMemcpyAsync(L.pinSizes[b], L.dSizes, nu * sizeof(uint32_t), hipMemcpyDeviceToHost, s) == hipSuccess;
                    ok = ok && hipEventRecord(L.ev[3], s) == hipSuccess;
                    ok = ok && hipStreamSynchronize(s) == hipSuccess;
                    if (!ok) r = ZERR(ZE_GENERIC);
                }
                if (zhip_isError(r)) { fail(r, k); break; }
                {   float a = 0, bb = 0, c2 = 0;
                    (void)hipEventElapsedTime(&a, L.evIn[b][0], L.evIn[b][1]); (void)hipEventElapsedTime(&bb, L.ev[1], L.ev[2]); (void)hipEventElapsedTime(&c2, L.ev[2], L.ev[3]);
                    tH2D += a * 1e-3; tK += bb * 1e-3; tD2H += c2 * 1e-3; }
                {   std::lock_guard<std::mutex> g(lm); inState[b] = 0; outState[b] = 1; outBytes[b] = r; }
                lcv.notify_all();
            }
        }
        feeder.join(); gatherer.join();
        {   std::lock_guard<std::mutex> g(gm);
            m->stages[0] += tIn; m->stages[1] += tH2D; m->stages[2] += tK; m->stages[3] += tD2H; m->stages[4] += tWait; m->stages[5] += tOut; m->stages[6] += (double)mine.size(); }
    };
    std::vector<std::thread> th;
    size_t const use = nChunks < nLanes ? nChunks : nLanes;
    for (size_t li = 1; li < use; li++) {
        try { th.emplace_back(lane_fn, li); }
        catch (...) {                                    // no thread for this lane: the call fails (the lanes that did start see firstErr at their next chunk and stop)
            {   std::lock_guard<std::mutex> g(gm);
                if (!firstErr) { firstErr = ZERR(ZE_memory_allocation); snprintf(m->err, sizeof(m->err), "no host thread for lane %zu", li); } }
            gcv.notify_all();
            break;
        }
    }
    lane_fn(0);
    for (auto& t : th) t.join();
    m->lastSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (firstErr) return firstErr;
    return off[nChunks];
}

// ONE input as the frame ZSTD_compress2 emits with ZSTD_c_nbWorkers >= 1 (see zhip_compress_frames_mt), its jobs spread over the lanes:
// a chunk is a run of consecutive jobs whose windows fit a lane's staging buffer; the lane copies [first window start, last job end)
// to the device, compresses the jobs there (a workgroup each) and hands the packed blocks to the same ordered gather — jobs are
// independent, so the copies of one chunk overlap the kernels of another, and several devices share one frame.  With ONE device, with
// the frame checksum on (one serial XXH64 over the whole input), for inputs up to 512 KB (no jobs) and for jobs larger than a staging
// buffer the call goes to a single context's zhip_compress_frames_mt instead.
size_t zhip_compress_frame_mt_multi(zhip_multi* m, void* dstv, size_t dstCapacity, const void* srcv, size_t srcSize,
                                    int level, const unsigned cparams[7], size_t jobSize, int overlapLog)
{
    std::lock_guard<std::mutex> lk(m->mu);
    if (cparams && !zhip::host_check_overrides(cparams)) return ZERR(ZE_parameter_outOfBound);
    if (overlapLog < 0 || overlapLog > 9) return ZERR(ZE_parameter_outOfBound);
    if (srcSize >= ((size_t)1 << 31)) { snprintf(m->err, sizeof(m->err), "inputs of 2 GiB and more are not implemented on device"); return ZERR(ZE_srcSize_wrong); }
    auto const t0 = std::chrono::steady_clock::now();
    uint8_t* const dst = (uint8_t*)dstv; const uint8_t* const src = (const uint8_t*)srcv;
    m->err[0] = 0;
    zhip::CParams cp;
    if (!zhip::host_get_cparams(level, srcSize, &cp, cparams)) return ZERR(ZE_parameter_unsupported);
    size_t section = srcSize ? srcSize : 1, overlap = 0;
    if (srcSize > zhip::MT_JOBSIZE_MIN) {
        section = zhip::host_mt_job_size(cp, jobSize); overlap = zhip::host_mt_overlap_size(cp, overlapLog);
        if (section < overlap) section = overlap;
    }
    zhip_multi_lane& L0 = m->lanes[0];
    size_t const room = L0.inCap > overlap + 80 ? L0.inCap - overlap - 80 : 0;
    size_t const perChunk = room / section;                               // jobs per chunk
    // One device: its single context is the faster host path — a job is a serial 2 MiB walk (45 ms at level 1), so the GPU wants some 500
    // jobs in flight, and chunks of a staging buffer's worth hold a few dozen (measured, 1 GiB, level 1: 14.2 GB/s PCIe-inclusive
    // against 3.0 GB/s in chunks of 31 jobs on 4 lanes).  The chunked lanes are for SEVERAL devices ($ZHIP_MULTI_FRAME_CHUNKED=1 forces them: tests).
    bool oneDevice = true;
    for (auto& L : m->lanes) oneDevice = oneDevice && L.device == L0.device;
    if (const char* e = getenv("ZHIP_MULTI_FRAME_CHUNKED")) { if (atoi(e) != 0) oneDevice = false; }
    if (oneDevice || srcSize <= zhip::MT_JOBSIZE_MIN || m->checksum || perChunk == 0 || (cp.strategy != ZHIP_STRAT_FAST && cp.strategy != ZHIP_STRAT_DFAST)) {
        unsigned long long const offs[2] = { 0, srcSize };
        size_t const need = (srcSize + section - 1) / section + 1;
        if (hipSetDevice(L0.device) != hipSuccess) return ZERR(ZE_GENERIC);
        if (!m->wide || m->wideUnits < need) {
            if (m->wide) zhip_destroy(m->wide);
            m->wideUnits = need < 64 ? 64 : need;
            m->wide = zhip_create(L0.device, m->wideUnits);
            if (!m->wide) { m->wideUnits = 0; return ZERR(ZE_memory_allocation); }
        }
        zhip_set_frame_checksum(m->wide, m->checksum);
        zhip_set_row_matcher(m->wide, m->rowMode);
        size_t const r = zhip_compress_frames_mt(m->wide, dstv, dstCapacity, srcv, offs, 1, level, cparams, jobSize, overlapLog, nullptr);
        if (zhip_isError(r)) snprintf(m->err, sizeof(m->err), "%s", zhip_last_error(m->wide));
        m->lastSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return r;
    }
    if (cp.windowLog < 17 && srcSize > ((size_t)1 << cp.windowLog)) return ZERR(ZE_parameter_unsupported);
    if (dstCapacity < zhip::host_compress_bound(srcSize)) return ZERR(ZE_dstSize_tooSmall);
    size_t const nJobs = (srcSize + section - 1) / section;
    size_t const nChunks = (nJobs + perChunk - 1) / perChunk, nLanes = m->lanes.size();
    std::vector<size_t> size(nChunks, 0), off(nChunks + 1, 0);
    std::vector<char> known(nChunks, 0);
    size_t placed = 0, firstErr = 0;
    std::mutex gm; std::condition_variable gcv;

    auto lane_fn = [&](size_t li) {
        zhip_multi_lane& L = m->lanes[li];
        if (hipSetDevice(L.device) != hipSuccess) { std::lock_guard<std::mutex> g(gm); if (!firstErr) firstErr = ZERR(ZE_GENERIC); gcv.notify_all(); return; }
        std::vector<zhip::ZhipJob> jobs; std::vector<uint32_t> lens;
        for (size_t k = li; k < nChunks; k += nLanes) {
            {   std::lock_guard<std::mutex> g(gm); if (firstErr) return; }
            size_t const j0 = k * perChunk, j1 = j0 + perChunk < nJobs ? j0 + perChunk : nJobs;
            jobs.clear(); lens.clear();
            for (size_t j = j0; j < j1; j++) {
                size_t const start = j * section, len = srcSize - start < section ? srcSize - start : section;
                zhip::ZhipJob jb;
                jb.start = (uint32_t)start; jb.prefixLen = (uint32_t)(j == 0 ? 0 : (section < overlap ? section : overlap));   // every earlier job is a full section
                jb.flags = (j == 0 ? ZHIP_JOB_FIRST : 0u) | (j + 1 == nJobs ? ZHIP_JOB_LAST : 0u);
                jb.ownHeader = zhip::frame_header_bytes_multi((uint32_t)len, cp.windowLog); jb.frameSize = srcSize; jb.frameIdx = 0; jb.pad0 = 0;
                jobs.push_back(jb); lens.push_back((uint32_t)len);
            }
            size_t const w0 = (size_t)jobs[0].start - jobs[0].prefixLen;
            size_t const end = (size_t)jobs.back().start + lens.back(), bytes = end - w0;
            // one byte of headroom in front: a job whose window starts at the chunk's first byte reads its position 1 = that byte
            memcpy(L.pinIn[0] + 16, src + w0, bytes);
            size_t r = 0;
            hipStream_t const s = L.ctx->stream;
            if (hipMemcpyAsync(L.dIn + 16, L.pinIn[0] + 16, bytes, hipMemcpyHostToDevice, s) != hipSuccess) r = ZERR(ZE_GENERIC);
            if (!r) r = frame_jobs_chunk_device(L.ctx, L.dOut, L.outCap, L.dIn + 16, w0, cp, jobs.data(), lens.data(), jobs.size(), s);
            if (!zhip_isError(r)) {
                bool ok = hipMemcpyAsync(L.pinOut[0], L.dOut, r, hipMemcpyDeviceToHost, s) == hipSuccess;
                ok = ok && hipStreamSynchronize(s) == hipSuccess;
                if (!ok) r = ZERR(ZE_GENERIC);
            }
            size_t myOff = 0;
            {   std::unique_lock<std::mutex> g(gm);
                if (zhip_isError(r)) { if (!firstErr) { firstErr = r; snprintf(m->err, sizeof(m->err), "jobs %zu..%zu on device %d: %s", j0, j1, L.device, zhip_last_error(L.ctx)); } gcv.notify_all(); return; }
                size[k] = r; known[k] = 1;
                while (placed < nChunks && known[placed]) { off[placed + 1] = off[placed] + size[placed]; placed++; }
                gcv.notify_all();
                gcv.wait(g, [&] { return placed > k || firstErr; });
                if (firstErr) return;
                myOff = off[k];
                if (myOff + r > dstCapacity) { firstErr = ZERR(ZE_dstSize_tooSmall); gcv.notify_all(); return; }
            }
            memcpy(dst + myOff, L.pinOut[0], r);
        }
    };
    std::vector<std::thread> th;
    size_t const use = nChunks < nLanes ? nChunks : nLanes;
    for (size_t li = 1; li < use; li++) {
        try { th.emplace_back(lane_fn, li); }
        catch (...) {                                    // no thread for this lane: the call fails (the lanes that did start see firstErr at their next chunk and stop)
            {   std::lock_guard<std::mutex> g(gm);
                if (!firstErr) { firstErr = ZERR(ZE_memory_allocation); snprintf(m->err, sizeof(m->err), "no host thread for lane %zu", li); } }
            gcv.notify_all();
            break;
        }
    }
    lane_fn(0);
    for (auto& t : th) t.join();
    m->lastSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (firstErr) return firstErr;
    return off[nChunks];
}

}  // extern "C"
