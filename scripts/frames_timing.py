#!/usr/bin/env python3
"""scripts/frames_timing.py — multi-block frames (zhip_compress_frames): kernel time of one 1 MiB frame and of a batch of frames
(one workgroup per frame: the batch is what fills the GPU).  Prints one JSON line per configuration."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
import zstd_amd as z

ctx = z.Context(max_units=1024)
LEVEL = int(os.environ.get("LEVEL", "1"))
for kind in ("datagen", "text"):
    for nf, size in ((1, 1 << 20), (256, 1 << 20), (1024, 1 << 20), (64, 16 << 20)):
        if kind == "datagen":
            base = z.datagen(size, 50, 1)
        else:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
            from _libs import text_like
            base = text_like(size, 1)
        bufs = [base] * nf
        for rep in range(2):
            outs = ctx.compress_frames(bufs, LEVEL)
        t = ctx.timing()
        print(json.dumps({"level": LEVEL, "kind": kind, "frames": nf, "frame_bytes": size, "timing_ms": t, "csize": len(outs[0]),
                          "GBps_kernel": round(nf * size / 1e6 / max(t["entropy_ms"] if "entropy_ms" in t else 1e-9, 1e-9), 3) if isinstance(t, dict) else None}))
