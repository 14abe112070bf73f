#!/usr/bin/env python3
"""scripts/big_frame_decode.py — ONE large frame (a job-pool frame of SIZE bytes) through the decoder: block-parallel since round 3
(zhip_decode_big.h; ZHIP_BIGFRAME_MIN=0 gives the one-workgroup walk of k_decode back: 0.26 GB/s)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import zstd_amd as z

N = int(os.environ.get("SIZE", str(256 << 20)))
a = z.datagen(N, 50, 1)
ctx = z.Context(max_units=600)
frame = ctx.compress_frames([a], 1, workers=1)[0]
d = z.DContext()
best = 1e9
for rep in range(2):
    t0 = time.perf_counter(); out = d.decompress(frame); best = min(best, time.perf_counter() - t0)
print(json.dumps({"bytes": N, "frame_bytes": len(frame), "decode_wall_s": round(best, 3), "timing": d.timing(), "GBps_wall": round(N / best / 1e9, 3), "GBps_device": round(N / d.timing()["decode_ms"] / 1e6, 3), "bigframe": d.last_bigframe(), "ok": out == a.tobytes()}))
