#!/usr/bin/env python3
"""scripts/frames_mt_timing.py — ONE large input as the reference's job-pool frame (zhip_compress_frames_mt, ZSTD_c_nbWorkers
semantics): the jobs of the frame are independent workgroups, so a single frame fills the GPU.  Prints one JSON line per
configuration: kernel time of the frame kernel (k_frame_fast over all jobs), checksum kernel, gather."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401  (HIP runtime first)
import zstd_amd as z

LEVEL = int(os.environ.get("LEVEL", "1"))
SIZE = int(os.environ.get("SIZE", str(256 << 20)))
ctx = z.Context(max_units=2048)
for kind in os.environ.get("KINDS", "datagen,text").split(","):
    if kind == "datagen":
        a = z.datagen(SIZE, 50, 1)
    else:
        from _libs import text_like
        a = np.tile(text_like(16 << 20, 1), SIZE // (16 << 20))
    for js in [int(x) for x in os.environ.get("JOBS", "0,524288,1048576,4194304").split(",")]:
        for ck in (False, True):
            if ck and (js != 0 or "JOBS" in os.environ):
                continue
            ctx.set_checksum(ck)
            for rep in range(2):
                outs = ctx.compress_frames([a], LEVEL, workers=1, job_size=js)
            t = ctx.timing()
            print(json.dumps({"level": LEVEL, "kind": kind, "bytes": int(a.size), "job_size": js, "checksum": ck, "timing_ms": t, "csize": len(outs[0]),
                              "jobs": int(ctx.stats()["units"]) if hasattr(ctx, "stats") else None}), flush=True)
    ctx.set_checksum(False)
