#!/usr/bin/env python3
"""tests/golden/make_dict_lazy_golden.py — dict_lazy_v1.json: SHA-256 of the REAL reference's frames for records compressed with a CDict at
the greedy / lazy / lazy2 levels (ZSTD_createCDict_advanced2 + ZSTD_CCtx_refCDict + ZSTD_compress2 on a fresh CCtx per record), with the
row matcher (default) and with the hash chain.  Oracle-only so far.  Run here: python tests/golden/make_dict_lazy_golden.py"""
import ctypes as C
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import load_oracle, load_ref, lazy_dict_cases, _buf, ERR

lo, lr = load_oracle(), load_ref()
lr.zref_compress_records_cdict_fresh.restype = C.c_size_t
lr.zref_compress_records_cdict_fresh.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
out = []
for name, d, recs in lazy_dict_cases(lo):
    src = np.concatenate(recs + [np.zeros(1, np.uint8)])
    sizes = (C.c_size_t * len(recs))(*[len(r) for r in recs])
    for level in (5, 6, 8, 10):
        for no_row in (0, 1):
            cap = sum(len(r) + (len(r) >> 7) + 256 for r in recs)
            dst = np.zeros(cap, dtype=np.uint8)
            k = lr.zref_compress_records_cdict_fresh(level, no_row, _buf(d), len(d), _buf(src), sizes, len(recs), _buf(dst), cap, None)
            assert k != ERR
            out.append({"case": name, "level": level, "noRow": no_row, "records": len(recs), "csize": int(k), "dst_sha256": hashlib.sha256(dst[:k].tobytes()).hexdigest()})
json.dump({"what": "facebook/zstd reference, CDict at lazy-strategy levels + refCDict + ZSTD_compress2, fresh CCtx per record, frames concatenated", "streams": out},
          open(os.path.join(HERE, "dict_lazy_v1.json"), "w"))
print(len(out), "streams")
