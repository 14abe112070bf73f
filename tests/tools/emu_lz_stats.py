#!/usr/bin/env python3
"""tests/tools/emu_lz_stats.py [bytes] [level] — no GPU: how the exact parse of a lazy-strategy frame serves its searches on the host SIMT emulator
(built with -DZHIP_LZ_STATS): searches, searches redone live from the live rows / by walking prev[], links followed by those walks, catch-up steps
of the live rows — for $ZHIP_LZ_RING = 0 / 1 and $ZHIP_LZ_PREDICT = 0 / 1, on datagen P50 and text-like input.  Every frame is compared with the oracle's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ZHIP_EMU_FLAGS"] = "-DZHIP_LZ_STATS"
import ctypes as C
import numpy as np
from _libs import load_oracle, load_emu, datagen, text_like, oracle_frame_params, emu_compress_frames_lazy, emu_compress_units, _buf, ERR

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
level = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lo, le = load_oracle(), load_emu()
lo.zo_set_row_matcher.argtypes = [C.c_int]; lo.zo_set_row_matcher(1)
cp = (C.c_uint * 7)(); assert lo.zo_get_cparams(level, n, cp) == 0
cpl = list(cp)
le.emu_lz_stats.argtypes = [C.c_void_p, C.c_int]
UNIT = 131072


def oracle_units(a):
    lo.zo_set_row_matcher(1)
    try:
        cap = lo.zo_compress_bound(UNIT) * (len(a) // UNIT + 1)
        dst = np.empty(cap, dtype=np.uint8)
        r = lo.zo_compress_chunks(level, UNIT, _buf(a), len(a), _buf(dst), cap, None, 0)
        assert r != ERR
        return dst[:r].tobytes()
    finally:
        lo.zo_set_row_matcher(0)


names = ["searches", "live:rows", "live:walk", "links", "catchup", "catchup-empty"]
if os.environ.get("UNITS", "1") != "0":
    for kind, a in (("datagen", datagen(lo, min(n, 4 * UNIT), 50, 7)), ("text", text_like(min(n, 2 * UNIT), 7))):
        want = oracle_units(a)
        bufs = [a[i:i + UNIT] for i in range(0, len(a), UNIT)]
        for ring in (0, 1):
            for pred in (0, 1):
                os.environ["ZHIP_LZ_RING"] = str(ring); os.environ["ZHIP_RH_PREDICT"] = str(pred)
                st = (C.c_ulonglong * 8)(); le.emu_lz_stats(st, 1)
                got = b"".join(emu_compress_units(le, lo, bufs, level, row=True))
                le.emu_lz_stats(st, 1)
                print(f"units {kind:8s} level {level} {len(a)} B ring {ring} predict {pred}: " + ", ".join(f"{k} {int(v)}" for k, v in zip(names, st)) +
                      ("   == oracle" if got == want else "   DIFFERS FROM THE ORACLE"), flush=True)
if os.environ.get("FRAMES", "1") == "0":
    sys.exit(0)
for kind, a in (("datagen", datagen(lo, n, 50, 7)), ("text", text_like(n, 7))):
    want = oracle_frame_params(lo, a, cp, True)
    for ring in (0, 1):
        for pred in (0, 1):
            os.environ["ZHIP_LZ_RING"] = str(ring); os.environ["ZHIP_LZ_PREDICT"] = str(pred)
            st = (C.c_ulonglong * 8)(); le.emu_lz_stats(st, 1)
            got = emu_compress_frames_lazy(le, lo, [a], [cpl], True, threads=1)[0]
            le.emu_lz_stats(st, 1)
            print(f"{kind:8s} level {level} {n} B ring {ring} predict {pred}: " + ", ".join(f"{k} {int(v)}" for k, v in zip(names, st)) +
                  ("   == oracle" if got == want else "   DIFFERS FROM THE ORACLE"), flush=True)
