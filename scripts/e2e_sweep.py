#!/usr/bin/env python3
"""scripts/e2e_sweep.py — GPU box, measurement helper: host-buffer path (zhip_compress_multi on one device) over lanes x chunk size; 1 GiB datagen, level 1"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
n = int(os.environ.get("MIB", "1024")) << 20
LANES = [int(x) for x in os.environ.get("LANES", "2,3").split(",")]
CUS = [int(x) for x in os.environ.get("CUS", "512,1024").split(",")]
res = []
for lanes in LANES:
    for cu in CUS:
        os.environ["ZHIP_MULTI_LANES"] = str(lanes)
        import zstd_amd
        host = zstd_amd.datagen(n, 50, seed=0, stream_mode=True)
        m = zstd_amd.MultiContext([0], chunk_units=cu)
        dst = np.empty(zstd_amd.compress_bound(n), dtype=np.uint8)
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter(); k = m.compress_into(dst, host, level=1); best = min(best, time.perf_counter() - t0)
        st = m.last_stages()
        m.close()
        print(json.dumps({"MiB": n >> 20, "lanes": lanes, "chunk_units": cu, "GBps": round(n / best / 1e9, 2), "csize": int(k), "stages_last_call": st}), flush=True)
