#!/usr/bin/env python3
"""scripts/occ_ab.py — A/B of register caps (more resident wavefronts) on the latency-bound kernels: runs the stage timings of
levels 1, 3, 5 (row hash) on datagen and text with the library named by $ZHIP_LIB (default: the built one).  One JSON line per run."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import zstd_amd as z
from _libs import text_like

N = int(os.environ.get("SIZE", str(512 << 20)))
ctx = z.Context(max_units=N // 131072 + 1)
for kind in ("datagen", "text"):
    a = z.datagen(N, 50, 1) if kind == "datagen" else np.tile(text_like(16 << 20, 1), N // (16 << 20))
    src = torch.from_numpy(a).cuda()
    dst = torch.empty(int(z.lib().zhip_compressBound(N, 131072)), dtype=torch.uint8, device="cuda")
    for level in [int(x) for x in os.environ.get("LEVELS", "1,3,5").split(",")]:
        best = None
        for rep in range(3):
            r = ctx.compress_device(dst.data_ptr(), dst.numel(), src.data_ptr(), N, level)
            t = ctx.timing()
            if best is None or t["total_ms"] < best["total_ms"]:
                best = t
        print(json.dumps({"lib": os.path.basename(os.environ.get("ZHIP_LIB", "libzstd_hip.so")), "kind": kind, "level": level, "bytes": N, "csize": int(r),
                          **{k: round(v, 2) for k, v in best.items()}}), flush=True)
