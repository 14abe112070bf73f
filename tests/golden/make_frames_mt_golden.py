#!/usr/bin/env python3
"""tests/golden/make_frames_mt_golden.py — frames_mt_v1.json: SHA-256 of the REAL reference's frame with ZSTD_c_nbWorkers = 1
(lib/compress/zstdmt_compress.c; ZSTD_c_jobSize / ZSTD_c_overlapLog / ZSTD_c_checksumFlag as listed in _libs.MT_MODES) for inputs
above 512 KB at the ZSTD_fast and ZSTD_dfast levels.  Run here: python tests/golden/make_frames_mt_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from _libs import load_oracle, load_ref, mt_frame_cases, ref_frame_mt, MT_MODES

lo, lr = load_oracle(), load_ref()
frames = []
for name, a in mt_frame_cases(lo):
    for level, js, ov, ck in MT_MODES:
        out = ref_frame_mt(lr, a, level, js, ov, bool(ck))
        frames.append({"case": name, "level": level, "jobSize": js, "overlapLog": ov, "checksum": ck,
                       "src_sha256": hashlib.sha256(a.tobytes()).hexdigest(), "csize": len(out), "dst_sha256": hashlib.sha256(out).hexdigest()})
json.dump({"what": "facebook/zstd reference, ZSTD_compress2 with ZSTD_c_nbWorkers=1 on a fresh CCtx, whole input -> one frame of jobs", "frames": frames},
          open(os.path.join(HERE, "frames_mt_v1.json"), "w"))
print(len(frames), "frames")
