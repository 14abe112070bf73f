#!/usr/bin/env python3
"""scripts/ab_queue.py — GPU box, measurement helper (not product): the ZSTD_fast stage in its launch forms — one workgroup per unit,
persistent queue, queue with a dispatch order (1 = estimated cost, 2 = the previous call's sequence counts: the bound an estimator can
reach), queue + the global-table co-kernel — on the three level-1 shapes; one JSON line per (form, shape) with a SHA-256 of the output.
usage: [ZHIP_LIB=...] python scripts/ab_queue.py MiB "q,o,g;q,o,g;..." [shapes]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import zstd_amd
from zstd_amd import workloads as W

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
forms = [tuple(int(x) for x in f.split(",")) for f in (sys.argv[2] if len(sys.argv) > 2 else "0,0,0;1,0,0;1,2,0;1,1,0;1,1,3").split(";")]
shapes = (sys.argv[3] if len(sys.argv) > 3 else "datagen,text,silesia").split(",")
level = int(os.environ.get("LEVEL", "1"))
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
    for (q, o, g) in forms:
        os.environ["ZHIP_FAST_QUEUE"] = str(q); os.environ["ZHIP_FAST_ORDER"] = str(o); os.environ["ZHIP_FAST_GWAVES"] = str(g)
        ctx = zstd_amd.Context(0, max_units=n // 131072 + 1)
        best = None
        for _ in range(4):
            r = ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, level, 131072)
            t = ctx.timing()
            if best is None or t["parse_ms"] < best["parse_ms"]:
                best = t
        out = dst[:r].cpu().numpy().tobytes()
        print(json.dumps({"lib": os.path.basename(zstd_amd.LIB_PATH), "shape": name, "MiB": mib, "queue": q, "order": o, "gwaves": g,
                          "parse_ms": round(best["parse_ms"], 3), "entropy_ms": round(best["entropy_ms"], 3),
                          "GBps": round(n / 1e6 / (best["parse_ms"] + best["entropy_ms"] + best["gather_ms"]), 2),
                          "sha": hashlib.sha256(out).hexdigest()[:16]}), flush=True)
        del ctx
