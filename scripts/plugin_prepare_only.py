#!/usr/bin/env python3
"""scripts/plugin_prepare_only.py [MiB] — the DEVICE part of the plugin_B1 leg alone: zhip_prepare_sequences on the headline's workload cut into the leg's 64 KB blocks
(H2D, k_parse_fast_q/_g, k_seq_compact, one packed copy back).  The leg itself (bench.py --leg plugin_B1) also runs the reference's entropy stage on 64 host threads,
under which rocprofv3 crashes (profiles/README_r06.md); this is what the counter passes of scripts/gpu_r6_profiles.sh run instead."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import zstd_amd
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
BLK = 65536
host = zstd_amd.datagen(mib << 20, 50, seed=0, stream_mode=True)
n = len(host) // BLK * BLK
a = np.ascontiguousarray(host[:n])
L = zstd_amd.lib()
ctx = zstd_amd.Context(0, max_units=n // BLK)
best = None
for _ in range(3):
    t0 = time.perf_counter()
    r = L.zhip_prepare_sequences(ctx._h, C.c_void_p(a.ctypes.data), n, BLK, 1)
    dt = time.perf_counter() - t0
    assert not L.zhip_isError(r)
    best = dt if best is None or dt < best else best
print(json.dumps({"blocks": n // BLK, "prepare_s": round(best, 4), "prepare_GBps": round(n / best / 1e9, 2), "device_parse_ms": ctx.timing()["parse_ms"]}))
