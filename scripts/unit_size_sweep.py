#!/usr/bin/env python3
"""scripts/unit_size_sweep.py — GPU box, measurement helper: the unit path at unit sizes 2 KB ... 128 KB (256 MiB of datagen / text, levels 1 / 3 / 5): device pipeline ms and the
host's wall time per call — a unit size at which either jumps is a cliff (per-unit host work, small-unit kernels)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zstd_amd
from zstd_amd import workloads as W

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
dev = torch.device("cuda", 0)
for kind in ("datagen", "text"):
    host = np.ascontiguousarray(zstd_amd.datagen(n, 50, seed=0, stream_mode=True) if kind == "datagen" else W.tile(W.text_corpus(32 << 20, seed=0), n))
    src = torch.empty(n + 64, dtype=torch.uint8, device=dev); src[:n].copy_(torch.from_numpy(host))
    for us in (2048, 8192, 8193, 32768, 131072):
        ctx = zstd_amd.Context(0, max_units=n // us + 1)
        cap = zstd_amd.compress_bound(n, us)
        dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
        for level in (1, 3, 5):
            best = None; wall = None
            for _ in range(3):
                t0 = time.time()
                r = ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, level, us)
                w = time.time() - t0
                t = ctx.timing()
                if best is None or t["total_ms"] < best["total_ms"]: best = t
                wall = w if wall is None or w < wall else wall
            print(json.dumps({"kind": kind, "unit": us, "level": level, "MiB": mib, "device_ms": round(best["total_ms"], 2), "parse_ms": round(best["parse_ms"], 2), "entropy_ms": round(best["entropy_ms"], 2),
                              "wall_ms": round(wall * 1e3, 2), "GBps_device": round(n / 1e6 / best["total_ms"], 2), "GBps_wall": round(n / 1e9 / wall, 2), "ratio": round(n / r, 3)}), flush=True)
        del ctx, dst
        torch.cuda.empty_cache()
