#!/usr/bin/env python3
"""scripts/frames_lazy_timing.py — multi-block frames of the lazy strategies (k_lz_links + k_lz_search + k_frame_lazy): device time of
batches of 1 MiB frames and of one large job-pool frame, per level.  Prints one JSON line per configuration; run it under
rocprofv3 --kernel-trace --stats for the split between the three kernels."""
import hashlib, json, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
import zstd_amd as z
from _libs import text_like

ctx = z.Context(max_units=1024)
REPS = int(os.environ.get("REPS", "2"))
LEVELS = [int(x) for x in os.environ.get("LEVELS", "5,7").split(",")]
for level in LEVELS:
    for kind in ("datagen", "text"):
        base = z.datagen(1 << 20, 50, 1) if kind == "datagen" else text_like(1 << 20, 1)
        for nf in [int(x) for x in os.environ.get("NFRAMES", "256,1024").split(",")]:
            bufs = [base] * nf
            for rep in range(REPS):
                outs = ctx.compress_frames(bufs, level)
            t = ctx.timing()
            print(json.dumps({"level": level, "kind": kind, "frames": nf, "frame_bytes": 1 << 20, "timing_ms": t, "csize": len(outs[0]), "sha": hashlib.sha256(outs[0]).hexdigest()[:16]}), flush=True)
    JP = int(os.environ.get("JOBPOOL_MIB", "256"))
    if JP <= 0:
        continue
    big = np.concatenate([z.datagen(JP << 18, 50, s) for s in range(4)])
    for rep in range(REPS):
        outs = ctx.compress_frames([big], level, workers=4)
    t = ctx.timing()
    print(json.dumps({"level": level, "kind": "datagen job-pool frame", "frames": 1, "frame_bytes": int(big.size), "timing_ms": t, "csize": len(outs[0]), "sha": hashlib.sha256(outs[0]).hexdigest()[:16]}), flush=True)
