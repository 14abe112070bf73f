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
    int rowMode = -1;                                   // zhip_set_row_matcher mode as last set (-1 = every context's own default)
    std::mutex mu;                                     // one call at a time
    char err[256] = {0};
    double lastSeconds = 0;
    zhip_ctx* wide = nullptr; size_t wideUnits = 0;    // job-pool frames that go to ONE context (checksum, jobs larger than a staging buffer): sized on demand
};

static void multi_free(zhip_multi_s* m)
{
    if (m->wide) zhip_destroy(m->wide);
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
    {   // every ordinal must exist BEFORE anything is touched: a failed hipSetDevice leaves a sticky "invalid device ordinal" behind that
        // the next, unrelated, launch check would report
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        for (int i = 0; i < nDevices; i++) if (devices[i] < 0 || devices[i] >= count) return nullptr;
    }
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
            memcpy(L.pinIn + 16, src + w0, bytes);
            size_t r = 0;
            hipStream_t const s = L.ctx->stream;
            if (hipMemcpyAsync(L.dIn + 16, L.pinIn + 16, bytes, hipMemcpyHostToDevice, s) != hipSuccess) r = ZERR(ZE_GENERIC);
            if (!r) r = frame_jobs_chunk_device(L.ctx, L.dOut, L.outCap, L.dIn + 16, w0, cp, jobs.data(), lens.data(), jobs.size(), s);
            if (!zhip_isError(r)) {
                bool ok = hipMemcpyAsync(L.pinOut, L.dOut, r, hipMemcpyDeviceToHost, s) == hipSuccess;
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
            memcpy(dst + myOff, L.pinOut, r);
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
