#!/usr/bin/env python3
"""tests/tools/emu_fuzz_lazy_units.py <seed> <seconds> — no GPU: 128 KB-class units of the strategies greedy / lazy / lazy2 with the ROW matcher on the host SIMT
emulator against the oracle, weighted towards long-match data (where positions are left out and searches go live): random effective parameters (hashLog 10-17,
searchLog 1-6 = rows of 16 / 32 / 64 entries, minMatch 3-7), the units' two-pass prediction on / off ($ZHIP_RH_PREDICT).
Prints BAD lines and saves the input under /tmp; `done <seed> <cases> bad <n>` at the end."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import _libs
from _libs import load_oracle, load_emu, datagen, text_like, emu_compress_units, _buf, ERR

seed, tmax = int(sys.argv[1]), float(sys.argv[2])
lo, le = load_oracle(), load_emu()
lo.zo_compress_unit_params.restype = C.c_size_t
lo.zo_compress_unit_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
lo.zo_set_row_matcher.argtypes = [C.c_int]
rng = np.random.default_rng(seed)
orig = _libs.make_units


def mk(n):
    kind = int(rng.integers(0, 6))
    s = int(rng.integers(0, 1 << 30))
    if kind <= 1:
        return datagen(lo, n, int(rng.choice([35, 50, 80, 95])), s)
    if kind == 2:                                                       # periodic with a few flips: matches of thousands of bytes
        per = int(rng.integers(300, 3000)); a = np.tile(rng.integers(0, 256, per, dtype=np.uint8), n // per + 1)[:n].copy()
        a[rng.integers(0, n, size=max(1, n // 3000))] ^= 0xFF; return a
    if kind == 3:                                                       # text then long-match data then noise: the rows start to matter in the middle
        k = n // 3; return np.concatenate([text_like(k, s), datagen(lo, k, 60, s), rng.integers(0, 256, n - 2 * k, dtype=np.uint8)])
    if kind == 4:
        a = datagen(lo, n, 50, s).copy(); k = n // 3; a[k:2 * k] = a[:k]; return a
    return text_like(n, s)


t0, cases, bad = time.time(), 0, 0
try:
    while time.time() - t0 < tmax:
        n = int(rng.choice([rng.integers(20000, 131072), 131072]))
        a = np.ascontiguousarray(mk(n))
        eff = [17, int(rng.integers(8, 18)), int(rng.integers(10, 18)), int(rng.integers(1, 7)), int(rng.integers(3, 8)), int(rng.choice([0, 2, 8, 16])), int(rng.choice([3, 4, 5]))]
        ring, pred = 0, int(rng.integers(0, 2))
        os.environ["ZHIP_RH_PREDICT"] = str(pred)
        lo.zo_set_row_matcher(1)
        cap = lo.zo_compress_bound(n) + 64
        o = np.zeros(cap, dtype=np.uint8)
        r = lo.zo_compress_unit_params(_buf(o), cap, _buf(a), n, (C.c_uint * 7)(*eff))
        assert r != ERR

        def mku(lo_, sizes, level_, unit=131072, row=False, eff=eff):
            units = orig(lo_, sizes, 1, unit, False)
            for f in units:
                f["windowLog"], f["chainLog"], f["hashLog"], f["searchLog"], f["minMatch"], f["targetLength"], f["strategy"] = eff
                f["litMode"] = 0
                f["rowLog"] = min(6, max(4, eff[3]))
            return units
        _libs.make_units = mku
        got = emu_compress_units(le, lo, [a], 1)[0]
        _libs.make_units = orig
        if got != o[:r].tobytes():
            bad += 1; print("BAD", seed, cases, n, eff, "ring", ring, "predict", pred, flush=True)
            np.save(f"/tmp/emu_fuzz_lazy_units_bad_{seed}_{cases}.npy", a)
        cases += 1
finally:
    _libs.make_units = orig
    lo.zo_set_row_matcher(0)
print("done", seed, cases, "bad", bad, flush=True)
