#!/usr/bin/env python3
"""tests/tools/jump_rounds.py [MiB] — how many k_bf_jump passes does a frame need on a machine that gives NO ordering between threads?
The emulator runs a pass thread after thread (a thread sees every shortcut made before it), the GPU gives anything in between, so the
worst case is the synchronous form: every entry reads the map as the previous pass left it.  Takes the copy map of a job-pool frame as
k_bf_build leaves it (emulator) and iterates it in numpy: with 1 hop beyond the first (reach x2 per pass: the form that was measured on
MI355X, 10 passes + 1 on the 1 GiB frame) and with up to 4 (reach x5: the form in the tree)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
from _libs import load_oracle, load_emu, datagen, text_like, oracle_frame_mt, _buf

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
lo, le = load_oracle(), load_emu()
le.emu_decode_big.restype = C.c_uint
le.emu_decode_big.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
le.emu_decode_big_want_map.argtypes = [C.c_void_p]
for kind in ("datagen", "text"):
    a = datagen(lo, mib << 20, 50, 1) if kind == "datagen" else text_like(mib << 20, 1)
    f = oracle_frame_mt(lo, a, 1, 0, 0, 0)
    src = np.frombuffer(f + b"\0" * 16, dtype=np.uint8).copy()
    dst = np.zeros(len(a) + 64, dtype=np.uint8); m0 = np.zeros(len(a) + 8, dtype=np.uint32)
    le.emu_decode_big_want_map(_buf(m0))
    osz, ck, rd = C.c_uint(0), C.c_uint(0), C.c_uint(0)
    st = le.emu_decode_big(_buf(src), len(f), _buf(dst), len(a), C.byref(osz), C.byref(ck), C.byref(rd), 0)
    le.emu_decode_big_want_map(None)
    assert st == 0 and dst[:len(a)].tobytes() == a.tobytes()
    m0 = m0[:len(a)].astype(np.int64)
    idx = np.arange(len(a), dtype=np.int64)
    res = {}
    for hops in (1, 4):
        m = m0.copy(); passes = 0
        while True:
            new = m[m]                                          # the first gather: map[map[i]]
            for _ in range(hops - 1):
                new = m[new]
            if (new == m).all():
                break
            m = new; passes += 1
        res[hops] = passes + 1                                  # + the pass that finds nothing to do
        assert (a[m] == a).all()
    print(kind, f"{mib} MiB", "matches' bytes", round(float((m0 != idx).mean()), 3), "passes (synchronous): reach x2", res[1], " reach x5", res[4], flush=True)
