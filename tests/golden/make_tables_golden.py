#!/usr/bin/env python3
"""tests/golden/make_tables_golden.py — digests of the REAL reference's stage-function answers for the histograms of
tests/_tables_cases.py (HUF_buildCTable_wksp + HUF_writeCTable_wksp; FSE_normalizeCount + FSE_writeNCount + FSE_buildCTable_wksp),
so that tests/test_gpu_tables.py is pinned on a box without /root/reference.  Run here: python tests/golden/make_tables_golden.py"""
import ctypes as C
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import load_ref
import _tables_cases as T

lr = load_ref()
out = {}
h = hashlib.sha256()
for c, m in T.huf_cases(seed=1, n=160):
    log, nb, hdr = T.ref_huf(lr, c, m, 11)
    hs = 0 if hdr is None else len(hdr)
    h.update(nb[: m + 1].tobytes()); h.update(bytes([log, hs])); h.update(hdr or b"")
out["huf_seed1_n160_max11"] = h.hexdigest()
lr.zref_fse_optimal_tablelog.restype = C.c_uint
lr.zref_fse_optimal_tablelog.argtypes = [C.c_uint, C.c_size_t, C.c_uint]
h = hashlib.sha256(); params = []
for c, total, maxSym, maxLog, lp in T.fse_cases(seed=2, n=240):
    tl = lr.zref_fse_optimal_tablelog(maxLog, total, maxSym)
    params.append([total, maxSym, tl, lp])
    rc, norm, hdr, tab = T.ref_fse(lr, c, total, maxSym, tl, lp)
    h.update(np.array([rc, len(hdr) if rc == 1 else 0], dtype=np.int32).tobytes())
    if rc == 1:
        st, df, db = tab
        h.update(norm[: maxSym + 1].tobytes()); h.update(hdr); h.update(st.tobytes()); h.update(db[: maxSym + 1].tobytes())
out["fse_seed2_n240"] = h.hexdigest()
out["fse_seed2_n240_params"] = params
json.dump(out, open(os.path.join(HERE, "tables_v1.json"), "w"))
print({k: v for k, v in out.items() if isinstance(v, str)})
