#!/usr/bin/env python3
"""scripts/shim_compress2_timing.py MiB level [device] — ZSTD_compress2 through libzstd_hipshim.so (the drop-in, include/zstd_hip_dropin.h) on the headline's workload, in a
process that holds nothing else (bench.py's end_to_end leg starts it: host-buffer timings depend on what a process did before, profiles/README_r05.md).  Prints one JSON line
with the rate and the SHA-256 of the stream."""
import ctypes as C, hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zstd_amd
from zstd_amd import build as zbuild
mib, level = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    os.environ["ZHIP_DEVICE"] = sys.argv[3]
host = zstd_amd.datagen(mib << 20, 50, seed=0, stream_mode=True)
zstd_amd.lib()
S = C.CDLL(zbuild.SHIM)
S.ZSTD_createCCtx.restype = C.c_void_p
S.ZSTD_freeCCtx.argtypes = [C.c_void_p]
S.ZSTD_CCtx_setParameter.restype = C.c_size_t; S.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
S.ZSTD_compress2.restype = C.c_size_t; S.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
S.ZSTD_compressBound.restype = C.c_size_t; S.ZSTD_compressBound.argtypes = [C.c_size_t]
S.ZSTD_isError.restype = C.c_uint; S.ZSTD_isError.argtypes = [C.c_size_t]
cap = S.ZSTD_compressBound(len(host))
dst = np.empty(cap, dtype=np.uint8)
cc = S.ZSTD_createCCtx(); S.ZSTD_CCtx_setParameter(cc, 100, level)
best, k = 1e9, 0
for _ in range(5):
    t0 = time.perf_counter()
    k = S.ZSTD_compress2(cc, dst.ctypes.data_as(C.c_void_p), cap, host.ctypes.data_as(C.c_void_p), len(host))
    best = min(best, time.perf_counter() - t0)
ok = not S.ZSTD_isError(k)
S.ZSTD_freeCCtx(cc)
print(json.dumps({"value": round(len(host) / best / 1e6, 1) if ok else None, "unit": "MB/s", "best_of": 5, "bytes": int(k) if ok else None,
                  "sha256": hashlib.sha256(dst[:k].tobytes()).hexdigest() if ok else None}))
