#!/usr/bin/env python3
"""tests/golden/make_frames_lazy_golden.py — frames_lazy_v1.json: SHA-256 of the REAL reference's single multi-block frame
(ZSTD_compress2 on a fresh CCtx, whole input in one call) at the greedy / lazy / lazy2 levels, with its default row-hash matcher
and with ZSTD_c_useRowMatchFinder = disable.  Oracle-only so far: the device has no frame kernel for these strategies yet
(DESIGN.md §9).  Run here: python tests/golden/make_frames_lazy_golden.py"""
import ctypes as C
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import load_oracle, load_ref, lazy_frame_cases, LAZY_FRAME_MODES, _buf, ERR

lo, lr = load_oracle(), load_ref()
lr.zref_compress_chunks_level_params.restype = C.c_size_t
lr.zref_compress_chunks_level_params.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
frames = []
for name, a in lazy_frame_cases(lo):
    for level, no_row in LAZY_FRAME_MODES:
        cp = (C.c_uint * 7)()
        assert lo.zo_get_cparams(level, len(a), cp) == 0 and cp[6] in (3, 4, 5)
        dst = np.zeros(len(a) + (len(a) >> 7) + 1024, dtype=np.uint8)
        r = lr.zref_compress_chunks_level_params(level, (C.c_int * 7)(0, 0, 0, 0, 0, 0, 0), no_row, len(a), _buf(a), len(a), _buf(dst), len(dst))
        assert r != ERR
        frames.append({"case": name, "level": level, "noRow": no_row, "strategy": int(cp[6]), "src_sha256": hashlib.sha256(a.tobytes()).hexdigest(),
                       "csize": int(r), "dst_sha256": hashlib.sha256(dst[:r].tobytes()).hexdigest()})
json.dump({"what": "facebook/zstd reference, ZSTD_compress2 on a fresh CCtx, whole input -> one multi-block frame, lazy strategies", "frames": frames},
          open(os.path.join(HERE, "frames_lazy_v1.json"), "w"))
print(len(frames), "frames")
