#!/usr/bin/env python3
"""Regenerates tests/golden/dict_v1.json with the REAL reference (oracle/_ref): for the committed ZDICT fixture and for a
raw-content dictionary, SHA-256 + size of what ZSTD_createCDict + ZSTD_CCtx_refCDict + ZSTD_compress2 emit per record
(levels 1, 3, 4), plus the SHA-256 of the records, so the dictionary path of the oracle is pinned where /root/reference and
oracle/_ref do not exist.  Run from the repo root:  python tests/golden/make_dict_golden.py"""
import ctypes as C, hashlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _libs import load_ref, _buf, ERR, text_like
from zstd_amd import workloads as W

lr = load_ref()
lr.zref_compress_records_cdict.restype = C.c_size_t
lr.zref_compress_records_cdict.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]


def records(seed):
    flat, offs = W.github_like_records(120, seed=seed)
    recs = [flat[int(offs[i]):int(offs[i + 1])].copy() for i in range(120)]
    t = text_like(40000, seed)
    recs += [np.zeros(0, np.uint8), recs[0][:6], recs[1][:7], recs[2][:8], recs[3][:9], recs[4][:100], np.concatenate(recs[5:12])[:8000], t[:3000], t[3000:3300]]
    return recs


out = []
zd = np.fromfile(os.path.join(ROOT, "tests", "golden", "github_like_110k.zdict"), dtype=np.uint8)
raw = W.github_like_records(100, seed=5)[0][:60000].copy()
for dname, d in (("zdict", zd), ("raw60000", raw)):
    for level in (1, 3, 4):
        recs = records(31)
        flat = np.concatenate(recs + [np.zeros(8, np.uint8)])
        sizes = (C.c_size_t * len(recs))(*[len(r) for r in recs])
        cap = sum(len(r) + 64 for r in recs) + 4096
        dst = np.zeros(cap, dtype=np.uint8)
        osz = (C.c_size_t * len(recs))()
        tot = lr.zref_compress_records_cdict(level, _buf(d), len(d), _buf(flat), sizes, len(recs), _buf(dst), cap, osz)
        assert tot != ERR
        out.append({"dict": dname, "dict_sha256": hashlib.sha256(d.tobytes()).hexdigest(), "level": level, "records_seed": 31,
                    "records_sha256": hashlib.sha256(flat[:-8].tobytes()).hexdigest(), "frame_sizes": [int(x) for x in osz],
                    "frames_sha256": hashlib.sha256(dst[:tot].tobytes()).hexdigest()})
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dict_v1.json")
json.dump({"reference": "facebook/zstd v1.5.6+dev @ /root/reference (2024-10-24): ZSTD_createCDict + ZSTD_CCtx_refCDict + ZSTD_compress2 per record",
           "cases": out}, open(path, "w"), indent=0)
print(len(out), "cases ->", path)
