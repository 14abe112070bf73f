#!/usr/bin/env python3
"""tests/golden/make_frames_golden.py — frames_v1.json: SHA-256 of the REAL reference's single multi-block frame
(ZSTD_compress2 on a fresh CCtx, whole input in one call) for inputs above 128 KB at the ZSTD_fast and ZSTD_dfast levels.
Run here: python tests/golden/make_frames_golden.py"""
import ctypes as C
import hashlib
import json
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import load_oracle, load_ref, frame_cases, _buf, ERR

lo, lr = load_oracle(), load_ref()
lr.zref_compress_frame.restype = C.c_size_t
lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
frames = []
for name, a in frame_cases(lo):
    for level in (1, 2, 3, 4, -1, -5):
        cp = (C.c_uint * 7)()
        assert lo.zo_get_cparams(level, len(a), cp) == 0
        if cp[6] not in (1, 2):
            continue                                   # strategies fast and dfast
        dst = np.zeros(len(a) + (len(a) >> 7) + 1024, dtype=np.uint8)
        r = lr.zref_compress_frame(level, _buf(a) if len(a) else None, len(a), _buf(dst), len(dst))
        assert r != ERR
        frames.append({"case": name, "level": level, "src_sha256": hashlib.sha256(a.tobytes()).hexdigest(), "csize": int(r),
                       "dst_sha256": hashlib.sha256(dst[:r].tobytes()).hexdigest()})
json.dump({"what": "facebook/zstd reference, ZSTD_compress2 on a fresh CCtx, whole input -> one multi-block frame", "frames": frames},
          open(os.path.join(HERE, "frames_v1.json"), "w"))
print(len(frames), "frames")
