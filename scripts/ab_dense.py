#!/usr/bin/env python3
"""scripts/ab_dense.py — GPU box, measurement helper (not product): the ZSTD_fast stage's two register budgets (zhip_kernels_parse.h: k_parse_fast_q / _g at three
waves per SIMD, _q4 / _g4 at four with seven global-table wavefronts per CU) on the level-1 shapes: $ZHIP_FAST_DENSE = 0 (never), 1 (always), unset (the cost-based
choice); one JSON line per (setting, shape) with a SHA-256 of the output.  usage: python scripts/ab_dense.py MiB [shapes]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zstd_amd
from zstd_amd import workloads as W

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
shapes = (sys.argv[2] if len(sys.argv) > 2 else "datagen,text,silesia").split(",")
n = mib << 20
dev = torch.device("cuda", 0)
cap = zstd_amd.compress_bound(n, 131072)
dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
gen = {
    "datagen": lambda: zstd_amd.datagen(n, 50, seed=0, stream_mode=True),
    "text": lambda: W.tile(W.text_corpus(64 << 20, seed=0), n),
    "silesia": lambda: W.tile(W.silesia_like(lambda size, P, seed: zstd_amd.datagen(size, P, seed=seed, stream_mode=False), seed=0), n),
}
for name in shapes:
    host = np.ascontiguousarray(gen[name]())
    src = torch.empty(n + 64, dtype=torch.uint8, device=dev); src[:n].copy_(torch.from_numpy(host))
    for dense in [None if x == "auto" else x for x in os.environ.get("AB_SET", "0,1,auto").split(",")]:
        if dense is None: os.environ.pop("ZHIP_FAST_DENSE", None)
        else: os.environ["ZHIP_FAST_DENSE"] = dense
        ctx = zstd_amd.Context(0, max_units=n // 131072 + 1)
        best = None
        for _ in range(5):
            r = ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, 1, 131072)
            t = ctx.timing()
            if best is None or t["parse_ms"] < best["parse_ms"]:
                best = t
        st = ctx.stats()
        out = dst[:r].cpu().numpy().tobytes()
        print(json.dumps({"shape": name, "MiB": mib, "lib": os.path.basename(zstd_amd.LIB_PATH), "gwaves_dense": os.environ.get("ZHIP_FAST_GWAVES_DENSE", "16-perCU"), "ZHIP_FAST_DENSE": dense if dense is not None else "auto", "parse_ms": round(best["parse_ms"], 3), "entropy_ms": round(best["entropy_ms"], 3),
                          "GBps": round(n / 1e6 / (best["parse_ms"] + best["entropy_ms"] + best["gather_ms"]), 2), "sequences_per_unit": round(st["sequences"] / st["units"]),
                          "sha": hashlib.sha256(out).hexdigest()[:16]}), flush=True)
        del ctx
