#!/usr/bin/env python3
"""tests/golden/make_params_golden.py — SHA-256 of the REAL reference's output for the explicit-parameter cases of
tests/test_gpu_params.py (ZSTD_CCtx_setParameter + ZSTD_compress2 per 128 KB chunk).  Run here: python tests/golden/make_params_golden.py"""
import ctypes as C
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import load_oracle, load_ref, datagen, text_like, _buf, ERR
import test_gpu_params as P

lo, lr = load_oracle(), load_ref()
lr.zref_compress_chunks_level_params.restype = C.c_size_t
lr.zref_compress_chunks_level_params.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
out = {}


def ref(a, level, cp):
    want = np.zeros(len(a) + len(a) // 64 + 4096, dtype=np.uint8)
    arr = (C.c_int * 7)(*cp)
    noRow = 1 if cp[6] in (3, 4, 5) or (cp[6] == 0 and level >= 5) else 0
    k = lr.zref_compress_chunks_level_params(level, arr, noRow, P.UNIT, _buf(a), len(a), _buf(want), len(want))
    assert k != ERR, (level, cp)
    return hashlib.sha256(want[:k].tobytes()).hexdigest()


for name, a in P.inputs(lo):
    for level, cp in P.PARAM_SETS:
        out[f"{name}|{level}|{','.join(map(str, cp))}"] = ref(a, level, cp)
a = datagen(lo, P.UNIT, 50, 33)
out["shim|datagen33|1|19,13,14,1,7,0,1"] = ref(a, 1, [19, 13, 14, 1, 7, 0, 1])
out["shim|datagen33|1|plain"] = ref(a, 1, [0] * 7)
json.dump(out, open(os.path.join(HERE, "params_v1.json"), "w"), indent=0)
print(len(out), "digests")
