#!/usr/bin/env python3
"""scripts/mt_multi_timing.py — PCIe-inclusive rate of ONE job-pool frame from pageable host memory: zhip_compress_frames_mt (one
context: blocking H2D / kernel / D2H) against zhip_compress_frame_mt_multi (lanes with pinned staging, copies overlapping kernels)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import zstd_amd as z

N = int(os.environ.get("SIZE", str(1 << 30)))
a = z.datagen(N, 50, 1)
import ctypes as C
L = z.lib()
L.zhip_compress_frames_mt.restype = C.c_size_t
L.zhip_compress_frames_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
offs = np.array([0, N], dtype=np.uint64)
for level, js in ((1, 0), (1, 524288), (3, 524288)):
    ctx = z.Context(max_units=2100)
    d1 = np.empty(z.compress_bound(N) + 64, dtype=np.uint8)
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        r = L.zhip_compress_frames_mt(ctx._h, d1.ctypes.data_as(C.c_void_p), d1.nbytes, a.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), 1, level, None, js, 0, None)
        best = min(best, time.perf_counter() - t0)
    out = d1[:r].tobytes()
    ctx.close()
    m = z.MultiContext([0])
    dst = np.empty(z.compress_bound(N) + 64, dtype=np.uint8)
    bm = 1e9
    for rep in range(4):
        k = m.compress_frame_mt_into(dst, a, level, job_size=js); bm = min(bm, m.last_seconds())
    same = dst[:k].tobytes() == out
    m.close()
    print(json.dumps({"level": level, "bytes": N, "job_size": js, "single_context_GBps": round(N / best / 1e9, 2), "multi_lane_GBps": round(N / bm / 1e9, 2),
                      "same_bytes": same, "csize": int(k)}), flush=True)
