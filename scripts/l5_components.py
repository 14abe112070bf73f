#!/usr/bin/env python3
"""scripts/l5_components.py [level] [MiB] — GPU box, measurement helper: the members of the Silesia-shaped mix (zstd_amd/workloads.py: silesia_like) one at a time through the
unit path at one level: where a level's time on the mix comes from.  Prints one JSON line per member (match-finder stage ms per GiB-equivalent, ratio)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zstd_amd
from zstd_amd import workloads as W

level = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 128
n = mib << 20
rng = np.random.default_rng(77)
dg = lambda P, seed: zstd_amd.datagen(n, P, seed=seed, stream_mode=False)
members = {
    "text_vocab8192": lambda: W.text_corpus(n, 1, vocab=8192),
    "datagen_P35": lambda: dg(35, 2),
    "text_vocab1024": lambda: W.text_corpus(n, 3, vocab=1024),
    "datagen_P60": lambda: dg(60, 4),
    "digits": lambda: rng.integers(0, 10, size=n, dtype=np.uint8) + 48,
    "datagen_P85": lambda: dg(85, 5),
    "random": lambda: rng.integers(0, 256, size=n, dtype=np.uint8),
    "runs_of_24": lambda: np.repeat(rng.integers(0, 256, size=n // 24 + 1, dtype=np.uint8), 24)[:n],
    "datagen_P50": lambda: dg(50, 6),
    # shapes outside the mix, each a known hard case of some match finder (EXTRA=1)
    "zeros": lambda: np.zeros(n, dtype=np.uint8),
    "period_2": lambda: np.tile(np.array([65, 66], dtype=np.uint8), n // 2),
    "period_7": lambda: np.tile(np.arange(7, dtype=np.uint8) + 48, n // 7 + 1)[:n],
    "runs_of_1000": lambda: np.repeat(rng.integers(0, 256, size=n // 1000 + 1, dtype=np.uint8), 1000)[:n],
    "two_symbols": lambda: rng.integers(0, 2, size=n, dtype=np.uint8) + 48,
    "one_line_x": lambda: np.tile(np.frombuffer(b"the quick brown fox jumps over the lazy dog and keeps on running through the field until dusk\n", dtype=np.uint8), n // 94 + 1)[:n],
    "le_u32_counter": lambda: np.arange(n // 4, dtype=np.uint32).view(np.uint8)[:n],
    "datagen_P98": lambda: dg(98, 8),
}
if os.environ.get("EXTRA") != "1":
    for k in ("zeros", "period_2", "period_7", "runs_of_1000", "two_symbols", "one_line_x", "le_u32_counter", "datagen_P98"):
        members.pop(k)
dev = torch.device("cuda", 0)
ctx = zstd_amd.Context(0, max_units=n // 131072 + 1)
if os.environ.get("PREDICT") is not None:
    ctx.set_prediction(units=int(os.environ["PREDICT"]))
cap = zstd_amd.compress_bound(n, 131072)
dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
src = torch.empty(n + 64, dtype=torch.uint8, device=dev)
for name, gen in members.items():
    host = np.ascontiguousarray(gen(), dtype=np.uint8)
    src[:n].copy_(torch.from_numpy(host))
    best = None
    for _ in range(2):
        r = ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, level, 131072)
        t = ctx.timing()
        if best is None or t["parse_ms"] < best["parse_ms"]:
            best = dict(t, hc={k: round(v, 2) for k, v in ctx.hc_timing().items()} if level >= 5 else None)
    print(json.dumps({"member": name, "level": level, "MiB": mib, "parse_ms": round(best["parse_ms"], 2), "parse_ms_per_GiB": round(best["parse_ms"] * 1024 / mib, 1),
                      "hc_ms": best.get("hc"), "ratio": round(n / r, 3)}), flush=True)
