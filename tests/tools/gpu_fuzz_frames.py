#!/usr/bin/env python3
"""tests/tools/gpu_fuzz_frames.py <seed> <trials> — on the GPU box: random inputs x explicit parameters x job sizes x overlaps through
zhip_compress_frames / zhip_compress_frames_mt against the oracle (which the CPU tests pin to the reference on the same generator);
real concurrency, which the host emulator cannot show."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import torch  # noqa: F401
import zstd_amd as z
from _libs import load_oracle, oracle_frame_mt, _buf
from test_oracle_vs_reference import mt_explicit_cases

lo = load_oracle()
lo.zo_compress_frame_params.restype = C.c_size_t
lo.zo_compress_frame_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
lo.zo_frame_bound.restype = C.c_size_t
lo.zo_frame_bound.argtypes = [C.c_size_t]
seed, trials = int(sys.argv[1]), int(sys.argv[2])
ctx = z.Context(max_units=32)
bad = n = 0
for a, level, req, eff, js, ov, ck in mt_explicit_cases(lo, trials, seed):
    ctx.set_checksum(ck)
    got = ctx.compress_frames([a], level, cparams=req, workers=1, job_size=js, overlap_log=ov)[0]
    ok = got == oracle_frame_mt(lo, a, level, js, ov, ck, cp=eff)
    ctx.set_checksum(False)
    cap = lo.zo_frame_bound(len(a)); o = np.zeros(cap, dtype=np.uint8)
    r = lo.zo_compress_frame_params(_buf(o), cap, _buf(a), len(a), eff)
    ok2 = ctx.compress_frames([a], level, cparams=req)[0] == o[:r].tobytes()
    n += 1
    if not (ok and ok2):
        bad += 1
        print("MISMATCH", len(a), level, req, list(eff), js, ov, ck, "jobs", ok, "frame", ok2, flush=True)
print("gpu fuzz: cases", n, "bad", bad)
