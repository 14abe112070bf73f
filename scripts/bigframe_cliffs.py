#!/usr/bin/env python3
"""scripts/bigframe_cliffs.py — GPU box, measurement helper: ONE large frame per shape (job-pool frame at level 1 / 3) through the job-pool compressor and the block-parallel
decoder (k_bf_*): compress kernel ms, decode ms, and whether the decoder stayed block-parallel.  A shape that takes many times the others' time is a cliff."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
import zstd_amd as z
from zstd_amd import workloads as W

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = mib << 20
rng = np.random.default_rng(9)
shapes = {
    "datagen_P50": lambda: z.datagen(n, 50, seed=1, stream_mode=False),
    "text": lambda: W.text_corpus(n, 1, vocab=4096),
    "zeros": lambda: np.zeros(n, dtype=np.uint8),
    "period_7": lambda: np.tile(np.arange(7, dtype=np.uint8) + 48, n // 7 + 1)[:n],
    "runs_of_24": lambda: np.repeat(rng.integers(0, 256, size=n // 24 + 1, dtype=np.uint8), 24)[:n],
    "runs_of_1000": lambda: np.repeat(rng.integers(0, 256, size=n // 1000 + 1, dtype=np.uint8), 1000)[:n],
    "two_symbols": lambda: rng.integers(0, 2, size=n, dtype=np.uint8) + 48,
    "random": lambda: rng.integers(0, 256, size=n, dtype=np.uint8),
    "one_line_x": lambda: np.tile(np.frombuffer(b"the quick brown fox jumps over the lazy dog and keeps on running through the field until dusk\n", dtype=np.uint8), n // 94 + 1)[:n],
    "period_300k": lambda: np.tile(rng.integers(0, 256, size=300000, dtype=np.uint8), n // 300000 + 1)[:n],
}
ctx = z.Context(0, max_units=max(64, n >> 19))
d = z.DContext()
for level in [int(a) for a in sys.argv[2:]] or [1, 3]:
    for name, gen in shapes.items():
        a = np.ascontiguousarray(gen(), dtype=np.uint8)
        t0 = time.time()
        comp = ctx.compress_frames([a], level, workers=4)[0]
        cw = time.time() - t0
        ct = ctx.timing()
        best = None
        for _ in range(2):
            t0 = time.time(); out = d.decompress(comp, capacity=n); w = time.time() - t0
            t = d.timing()["decode_ms"]
            best = t if best is None or t < best else best
        ok = out == a.tobytes()
        print(json.dumps({"shape": name, "level": level, "MiB": mib, "compress_kernel_ms": round(ct["entropy_ms"] + ct["parse_ms"], 2), "compress_wall_s": round(cw, 2),
                          "decode_ms": round(best, 2), "decode_wall_s": round(w, 2), "ok": bool(ok), "ratio": round(n / len(comp), 1), **d.last_bigframe()}), flush=True)
