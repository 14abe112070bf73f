#!/usr/bin/env python3
"""scripts/decode_cliffs.py — GPU box, measurement helper: the batch decoder (k_decode: 128 KB frames made by the unit path at level 1 and 5) on hard shapes: decode ms per
GiB-equivalent per shape.  A shape that takes many times the others' time is a cliff (overlapping match copies, RLE blocks, long literal runs ...)."""
import json, os, sys
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
    "period_2": lambda: np.tile(np.array([65, 66], dtype=np.uint8), n // 2),
    "period_7": lambda: np.tile(np.arange(7, dtype=np.uint8) + 48, n // 7 + 1)[:n],
    "runs_of_24": lambda: np.repeat(rng.integers(0, 256, size=n // 24 + 1, dtype=np.uint8), 24)[:n],
    "runs_of_1000": lambda: np.repeat(rng.integers(0, 256, size=n // 1000 + 1, dtype=np.uint8), 1000)[:n],
    "two_symbols": lambda: rng.integers(0, 2, size=n, dtype=np.uint8) + 48,
    "digits": lambda: rng.integers(0, 10, size=n, dtype=np.uint8) + 48,
    "random": lambda: rng.integers(0, 256, size=n, dtype=np.uint8),
    "one_line_x": lambda: np.tile(np.frombuffer(b"the quick brown fox jumps over the lazy dog and keeps on running through the field until dusk\n", dtype=np.uint8), n // 94 + 1)[:n],
}
ctx = z.Context(0, max_units=n // 131072 + 1)
d = z.DContext()
for level in (1, 5):
    for name, gen in shapes.items():
        a = np.ascontiguousarray(gen(), dtype=np.uint8)
        comp = ctx.compress(a, level=level)
        best = None
        for _ in range(3):
            out = d.decompress(comp, capacity=n)
            t = d.timing()["decode_ms"]
            best = t if best is None or t < best else best
        assert out == a.tobytes(), (name, level)
        print(json.dumps({"shape": name, "level": level, "MiB": mib, "decode_ms": round(best, 3), "decode_ms_per_GiB": round(best * 1024 / mib, 1), "ratio": round(n / len(comp), 2)}), flush=True)
