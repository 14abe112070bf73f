#!/usr/bin/env python3
"""scripts/shape_times.py — GPU box: level-1 match finder time per data shape of the Silesia-shaped mix (256 MiB each)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zstd_amd
from zstd_amd import workloads as W
n = 256 << 20
rng = np.random.default_rng(5)
shapes = {
    "text_v8192": lambda: W.tile(W.text_corpus(32 << 20, 1, vocab=8192), n),
    "text_v1024": lambda: W.tile(W.text_corpus(32 << 20, 3, vocab=1024), n),
    "datagen_P35": lambda: zstd_amd.datagen(n, 35, seed=2, stream_mode=False),
    "datagen_P60": lambda: zstd_amd.datagen(n, 60, seed=4, stream_mode=False),
    "datagen_P85": lambda: zstd_amd.datagen(n, 85, seed=5, stream_mode=False),
    "digits": lambda: (rng.integers(0, 10, size=n, dtype=np.uint8) + 48),
    "random": lambda: rng.integers(0, 256, size=n, dtype=np.uint8),
    "runs24": lambda: np.repeat(rng.integers(0, 256, size=n // 24 + 1, dtype=np.uint8), 24)[:n],
}
dev = torch.device("cuda", 0)
ctx = zstd_amd.Context(0, max_units=n // 131072)
cap = zstd_amd.compress_bound(n, 131072)
dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
for name, gen in shapes.items():
    host = np.ascontiguousarray(gen())
    src = torch.empty(n + 64, dtype=torch.uint8, device=dev); src[:n].copy_(torch.from_numpy(host))
    best = (1e9, 0)
    for _ in range(3):
        r = ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, 1, 131072)
        t = ctx.timing()
        if t["parse_ms"] < best[0]: best = (t["parse_ms"], t["entropy_ms"])
    print(f"{name:12s} parse {best[0]:8.2f} ms  entropy {best[1]:6.2f} ms  ratio {n / r:6.3f}  -> {n / 1e6 / (best[0] + best[1]):8.1f} GB/s")
