#!/usr/bin/env python3
"""tests/tools/gpu_fuzz_units.py <seed> <trials> — on the GPU box: structured-random data cut into units of random sizes, random explicit
parameters (every strategy up to lazy2, row matcher on / off), through zhip_compress_params against the oracle unit by unit,
and back through the device decoder."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import torch  # noqa: F401
import zstd_amd as z
from _libs import load_oracle, _buf, ERR, datagen
from test_fuzz_emu import gen, _explicit

lo = load_oracle()
lo.zo_compress_unit_params.restype = C.c_size_t
lo.zo_compress_unit_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
L = z.lib()
L.zhip_compress_params.restype = C.c_size_t
L.zhip_compress_params.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
seed, trials = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
ctx = z.Context(max_units=256)
dctx = z.DContext()
bad = cases = units = 0
for t in range(trials):
    unit = int(rng.choice([131072, 65536, 20000, 4096, 100000]))
    n = int(rng.integers(unit, 24 * unit))
    a = gen(rng, n) if t % 3 else np.concatenate([gen(rng, n // 2), datagen(lo, n - n // 2, int(rng.integers(5, 95)), t)])
    level = int(rng.choice([1, 3, 5, 6, 7, -3]))
    req = [int(rng.choice([0, 0, 17, 18])), int(rng.choice([0, 0, 8, 12, 15, 16])), int(rng.choice([0, 0, 8, 11, 13, 15, 17])),
           int(rng.choice([0, 0, 1, 2, 4, 5, 6])), int(rng.choice([0, 0, 3, 4, 5, 6, 7])), int(rng.choice([0, 0, 1, 4, 16, 64])), int(rng.choice([0, 0, 1, 2, 3, 4, 5]))]
    no_row = int(rng.integers(0, 2))
    os.environ["ZHIP_ROW_MATCHER"] = "disable" if no_row else "auto"
    cap = z.compress_bound(n, unit)
    dst = np.empty(cap, dtype=np.uint8); sizes = np.zeros(n // unit + 2, dtype=np.uint64)
    if hasattr(ctx, "set_row_matcher"):
        ctx.set_row_matcher(0 if not no_row else 2)
    r = L.zhip_compress_params(ctx._h, dst.ctypes.data_as(C.c_void_p), cap, a.ctypes.data_as(C.c_void_p), n, level, (C.c_uint * 7)(*req), unit, sizes.ctypes.data_as(C.c_void_p))
    if L.zhip_isError(r):
        continue                                             # parameters the device does not run
    cases += 1
    if dctx.decompress(dst[:r].tobytes()) != a.tobytes():    # the device decoder on the same frames
        bad += 1
        print("DECODE MISMATCH trial", t, level, req, flush=True)
    pos = 0
    for k in range(-(-n // unit)):
        u = a[k * unit: (k + 1) * unit]
        eff = _explicit(level, len(u), req)
        row = 3 <= eff[6] <= 5 and eff[0] > 14 and not no_row
        lo.zo_set_row_matcher(1 if row else 0)
        o = np.zeros(lo.zo_compress_bound(len(u)) + 64, dtype=np.uint8)
        rr = lo.zo_compress_unit_params(_buf(o), len(o), _buf(u), len(u), eff)
        got = dst[pos: pos + int(sizes[k])].tobytes(); pos += int(sizes[k])
        units += 1
        if rr == ERR or got != o[:rr].tobytes():
            bad += 1
            print("MISMATCH trial", t, "unit", k, len(u), level, req, list(eff), "noRow", no_row, flush=True)
            break
lo.zo_set_row_matcher(0)
print("gpu unit fuzz: calls", cases, "units", units, "bad", bad)
