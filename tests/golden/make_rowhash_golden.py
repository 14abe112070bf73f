#!/usr/bin/env python3
"""tests/golden/make_rowhash_golden.py — units_v3_rowhash.json: SHA-256 of the REAL reference's frames at levels 5-10 with its
DEFAULT match finder (row hash when windowLog > 14), one FRESH CCtx per unit (ZSTD_createCCtx + ZSTD_compress2: the row matcher's
hash salt is a per-CCtx-history value, zstd_compress.c:1964-1975).  Run here: python tests/golden/make_rowhash_golden.py"""
import ctypes as C
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import load_oracle, load_ref, corpus_cases, _buf, ERR

lo, lr = load_oracle(), load_ref()
lr.zref_compress_frame.restype = C.c_size_t
lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
units = []
for n in (131072, 100001, 40000, 20000):
    for name, a in corpus_cases(lo, sizes=(n,), seeds=(0,)):
        for level in (5, 6, 7, 8, 9, 10):
            cp = (C.c_uint * 7)()
            if lo.zo_get_cparams(level, n, cp) != 0:
                continue
            dst = np.zeros(n + 1024, dtype=np.uint8)
            r = lr.zref_compress_frame(level, _buf(a), n, _buf(dst), len(dst))
            assert r != ERR
            units.append({"case": name, "level": level, "src_sha256": hashlib.sha256(a.tobytes()).hexdigest(), "csize": int(r),
                          "dst_sha256": hashlib.sha256(dst[:r].tobytes()).hexdigest()})
json.dump({"what": "facebook/zstd reference, ZSTD_compress2 on a fresh CCtx per unit, default match finder (row hash)", "units": units},
          open(os.path.join(HERE, "units_v3_rowhash.json"), "w"))
print(len(units), "units")
