#!/usr/bin/env python3
"""scripts/ab_parse.py — GPU box, measurement helper (not product): times the pipeline of one library build ($ZHIP_LIB, default the
product build) on the three level-1 input shapes and prints one JSON line per shape with a SHA-256 of the compressed stream, so that
two builds can be compared for speed AND for identical bytes.   usage: [ZHIP_LIB=...] python scripts/ab_parse.py [level] [shapes,...] [MiB]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zstd_amd
from zstd_amd import workloads as W

level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
shapes = (sys.argv[2] if len(sys.argv) > 2 else "datagen,text,silesia").split(",")
mib = int(sys.argv[3]) if len(sys.argv) > 3 else 512
n = mib << 20
dev = torch.device("cuda", 0)
ctx = zstd_amd.Context(0, max_units=n // 131072 + 1)
if os.environ.get("ROW") is not None:
    ctx.set_row_matcher(int(os.environ["ROW"]))
if os.environ.get("PREDICT") is not None:                       # the row matcher's two-pass prediction for units (off by default)
    ctx.set_prediction(units=int(os.environ["PREDICT"]))
cap = zstd_amd.compress_bound(n, 131072)
dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
gen = {
    "datagen": lambda: zstd_amd.datagen(n, 50, seed=0, stream_mode=True),
    "text": lambda: W.tile(W.text_corpus(64 << 20, seed=0), n),
    "silesia": lambda: W.tile(W.silesia_like(lambda size, P, seed: zstd_amd.datagen(size, P, seed=seed, stream_mode=False), seed=0), n),
    # the bench's own leg (bench.py --workload silesia --copies 4): 4 copies of the mix, 6 468 units — not a whole number of rounds of the resident wavefronts
    "silesia4": lambda: np.tile(W.silesia_like(lambda size, P, seed: zstd_amd.datagen(size, P, seed=seed, stream_mode=False), seed=0), 4),
    "text1e9": lambda: W.tile(W.text_corpus(64 << 20, seed=0), 1000000000),
}
nmax = n
for name in shapes:
    n = nmax
    host = np.ascontiguousarray(gen[name]())
    n = min(len(host), nmax)                                    # (the generators read `n`: every shape starts from the asked size)
    src = torch.empty(n + 64, dtype=torch.uint8, device=dev); src[:n].copy_(torch.from_numpy(host[:n]))
    best = None
    for _ in range(4):
        r = ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, level, 131072)
        t = ctx.timing()
        if best is None or t["parse_ms"] + t["entropy_ms"] < best["parse_ms"] + best["entropy_ms"]:
            best = t
            if level >= 5:
                best = dict(t, hc={k: round(v, 2) for k, v in ctx.hc_timing().items()})
    out = dst[:r].cpu().numpy().tobytes()
    print(json.dumps({"lib": os.path.basename(zstd_amd.LIB_PATH), "shape": name, "level": level, "MiB": mib, "parse_ms": round(best["parse_ms"], 3),
                      "entropy_ms": round(best["entropy_ms"], 3), "GBps": round(n / 1e6 / (best["parse_ms"] + best["entropy_ms"] + best["gather_ms"]), 2),
                      "hc_ms": best.get("hc"), "ratio": round(n / r, 4), "sha": hashlib.sha256(out).hexdigest()[:16]}), flush=True)
