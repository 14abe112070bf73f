#!/usr/bin/env python3
"""Regenerates tests/golden/units_v1.json with the REAL reference (oracle/_ref, built from /root/reference).

Each entry pins, for a seeded synthetic input and a level, the byte length and SHA-256 of what the reference's
ZSTD_compress2 emits for that input as ONE independent unit (frame), plus the SHA-256 of the input itself, so the
fixture also pins the input generators.  Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from _libs import load_oracle, load_ref, corpus_cases, _buf, ERR

lo, lr = load_oracle(), load_ref()
out = []
for level in (1, 3):
    for n in (131072, 100000, 16384, 5000, 300, 64, 7, 0):
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(0, 5)):
            cap = lr.zref_compress_bound(n) + 64
            dst = np.zeros(cap, dtype=np.uint8)
            r = lr.zref_compress_chunks(level, 1 << 17, _buf(a), n, _buf(dst), cap, None, 0)
            assert r != ERR
            out.append({"case": name, "level": level, "n": n,
                        "src_sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                        "csize": int(r), "dst_sha256": hashlib.sha256(dst[:r].tobytes()).hexdigest()})
# hash-chain strategies (greedy / lazy / lazy2, levels 5-7 of the <= 128 KB row and 4-6 of the <= 16 KB row) with
# ZSTD_c_useRowMatchFinder = ZSTD_ps_disable: separate file so that units_v1.json stays as it was
hc = []
for level in (5, 6, 7):
    for n in (131072, 100000, 16384, 5000, 300, 64, 7, 0):
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(0, 5)):
            cap = lr.zref_compress_bound(n) + 64
            dst = np.zeros(cap, dtype=np.uint8)
            r = lr.zref_compress_chunks_norow(level, 1 << 17, _buf(a), n, _buf(dst), cap, None, 0)
            assert r != ERR
            hc.append({"case": name, "level": level, "n": n,
                       "src_sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                       "csize": int(r), "dst_sha256": hashlib.sha256(dst[:r].tobytes()).hexdigest()})
path = os.path.join(os.path.dirname(__file__), "units_v2_hashchain.json")
json.dump({"reference": "facebook/zstd v1.5.6+dev @ /root/reference (2024-10-24), ZSTD_c_useRowMatchFinder=ZSTD_ps_disable",
           "units": hc}, open(path, "w"), indent=0)
print(len(hc), "entries ->", path)
path = os.path.join(os.path.dirname(__file__), "units_v1.json")
json.dump({"reference": "facebook/zstd v1.5.6+dev @ /root/reference (2024-10-24)", "units": out}, open(path, "w"), indent=0)
print(len(out), "entries ->", path)
