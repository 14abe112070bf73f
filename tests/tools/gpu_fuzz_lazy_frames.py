#!/usr/bin/env python3
"""tests/tools/gpu_fuzz_lazy_frames.py <seed> <seconds> — on the GPU box: multi-block frames and job-pool frames of the strategies greedy / lazy / lazy2
(zhip_frame_lazy.h) through zhip_compress_frames / zhip_compress_frames_mt with random explicit parameters (windowLog 17-21, table logs 8-20,
searchLog 1-6: rows of 16 / 32 / 64 entries), row matcher or hash chain, the two-pass prediction on or off, against the oracle (pinned to the reference on
the same parameter domain by tests/test_oracle_vs_reference.py) — the GPU twin of `emu_fuzz_frames.py lazyframes | lazyjobs`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import torch  # noqa: F401
import zstd_amd as z
from _libs import load_oracle, datagen, text_like, oracle_frame_mt, oracle_frame_params

seed, tmax = int(sys.argv[1]), float(sys.argv[2])
lo = load_oracle()
lo.zo_set_row_matcher.argtypes = [C.c_int]
L = z.lib()
L.zhip_getCParams_explicit.restype = C.c_int
L.zhip_getCParams_explicit.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p]
rng = np.random.default_rng(seed)
ctx = z.Context(max_units=64)


def mk(n, kind, s, piece=400000):
    if kind == 0:
        return datagen(lo, n, int(rng.choice([20, 50, 80, 95])), s)
    if kind == 1:
        return text_like(n, s)
    if kind == 2:
        return rng.integers(0, 256, size=n, dtype=np.uint8)
    if kind == 3:                                                       # a third of the input repeats the first third: far matches
        a = datagen(lo, n, 50, s).copy(); k = n // 3; a[k:2 * k] = a[:k]; return a
    if kind == 4:                                                       # zeros with a few ones: RLE blocks, huge matches
        a = np.zeros(n, np.uint8); a[rng.integers(0, n, size=max(1, n // 5000))] = 1; return a
    parts, left = [], n
    while left > 0:
        m = min(left, int(rng.integers(100, piece))); parts.append(mk(m, int(rng.integers(0, 5)), s + len(parts))); left -= m
    return np.concatenate(parts)


def eff_of(n, req):
    eff = (C.c_uint * 7)()
    return eff if L.zhip_getCParams_explicit(5, n, (C.c_uint * 7)(*req), eff) == 0 and 3 <= eff[6] <= 5 and eff[0] >= 17 else None


t0 = time.time(); cases = bad = 0
while time.time() - t0 < tmax:
    wl = int(rng.integers(17, 22))
    req = [wl, int(rng.integers(8, wl + 1)), int(rng.integers(10, min(wl + 1, 20) + 1)), int(rng.integers(1, 7)), int(rng.integers(3, 8)), int(rng.choice([0, 2, 8, 16])), int(rng.choice([3, 4, 5]))]
    row = int(rng.integers(0, 2)); pred = int(rng.integers(0, 2)); ck = bool(rng.integers(0, 2))
    ctx.set_row_matcher(0 if row else 2); ctx.set_prediction(frames=pred); ctx.set_checksum(ck)
    if cases % 2 == 0:
        bufs = [np.ascontiguousarray(mk(int(rng.choice([rng.integers(131073, 140000), rng.integers(131073, 900000)])),
                                        int(rng.integers(0, 6)), int(rng.integers(0, 1 << 30)), 150000)) for _ in range(int(rng.integers(1, 4)))]
        effs = [eff_of(len(b), req) for b in bufs]
        if any(e is None for e in effs):
            continue
        ctx.set_checksum(False)
        outs = ctx.compress_frames(bufs, 5, cparams=req)
        for b, e, g in zip(bufs, effs, outs):
            if g != oracle_frame_params(lo, b, e, 1 if (row and e[0] > 14) else 0):
                bad += 1; print("BAD frames", seed, cases, len(b), req, list(e), row, pred, flush=True)
    else:
        a = np.ascontiguousarray(mk(int(rng.integers(524289, 3_000_000)), int(rng.integers(0, 6)), int(rng.integers(0, 1 << 30)), 250000))
        e = eff_of(len(a), req)
        if e is None:
            continue
        js = int(rng.choice([0, 524288, 524288, int(rng.integers(524288, 900000))])); ov = int(rng.choice([0, 0, 1, 3, 6, 8, 9]))
        got = ctx.compress_frames([a], 5, cparams=req, workers=2, job_size=js, overlap_log=ov)[0]
        lo.zo_set_row_matcher(1 if (row and e[0] > 14) else 0)
        try:
            want = oracle_frame_mt(lo, a, 5, js, ov, ck, cp=e)
        finally:
            lo.zo_set_row_matcher(0)
        if got != bytes(want):
            bad += 1; print("BAD jobs", seed, cases, len(a), req, list(e), row, pred, js, ov, ck, flush=True)
    cases += 1
print("gpu lazy-frame fuzz: seed", seed, "cases", cases, "bad", bad, flush=True)
