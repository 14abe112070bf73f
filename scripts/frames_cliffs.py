#!/usr/bin/env python3
"""scripts/frames_cliffs.py — GPU box, measurement helper: multi-block frames (zhip_compress_frames) of hard shapes at several levels; prints the frame kernels' time per shape and level
(64 frames of 1 MiB each: one round of the resident workgroups, i.e. the latency of one frame)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import zstd_amd as z

n = 1 << 20
rng = np.random.default_rng(5)
shapes = {
    "datagen_P50": lambda: z.datagen(n, 50, 1),
    "zeros": lambda: np.zeros(n, dtype=np.uint8),
    "runs_of_24": lambda: np.repeat(rng.integers(0, 256, size=n // 24 + 1, dtype=np.uint8), 24)[:n],
    "runs_of_1000": lambda: np.repeat(rng.integers(0, 256, size=n // 1000 + 1, dtype=np.uint8), 1000)[:n],
    "period_7": lambda: np.tile(np.arange(7, dtype=np.uint8) + 48, n // 7 + 1)[:n],
    "two_symbols": lambda: rng.integers(0, 2, size=n, dtype=np.uint8) + 48,
    "digits": lambda: rng.integers(0, 10, size=n, dtype=np.uint8) + 48,
    "random": lambda: rng.integers(0, 256, size=n, dtype=np.uint8),
}
ctx = z.Context(max_units=64)
for level in [int(a) for a in sys.argv[1:]] or [1, 3, 5, 7]:
    for name, gen in shapes.items():
        a = np.ascontiguousarray(gen(), dtype=np.uint8)
        for _ in range(2):
            outs = ctx.compress_frames([a] * 64, level)
        t = ctx.timing()
        print(json.dumps({"shape": name, "level": level, "frames": 64, "kernel_ms": round(t["entropy_ms"] + t["parse_ms"], 2), "csize": len(outs[0])}), flush=True)
