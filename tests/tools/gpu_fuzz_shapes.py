#!/usr/bin/env python3
"""tests/tools/gpu_fuzz_shapes.py <seed> <trials> — on the GPU box: DEGENERATE shapes (regular runs, short periods, two symbols, a repeated line with a few mutations, zeros with
noise islands, counters) cut into units of random sizes, at every level family (fast, dfast, greedy / lazy / lazy2 with both matchers), through the unit path against the oracle
unit by unit, through the frame path (one multi-block frame per buffer) against the oracle's frame, and back through the device decoder.  Round 6: the shapes that found the two
quadratic loops of the lazy parsers (profiles/README_r06.md) — this is their parity net."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import torch  # noqa: F401
import zstd_amd as z
from _libs import load_oracle, _buf, ERR, datagen, oracle_frame_params

lo = load_oracle()
lo.zo_compress_unit.restype = C.c_size_t
seed, trials = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)


def shape(rng, n):
    k = int(rng.integers(0, 8))
    if k == 0:
        r = int(rng.choice([2, 3, 5, 8, 24, 31, 64, 100, 385, 1000, 5000]))
        return np.repeat(rng.integers(0, int(rng.choice([2, 16, 256])), size=n // r + 1, dtype=np.uint8), r)[:n]
    if k == 1:
        p = int(rng.choice([1, 2, 3, 7, 16, 33, 255, 700, 4099, 70000]))
        return np.tile(rng.integers(0, 256, size=p, dtype=np.uint8), n // p + 1)[:n]
    if k == 2:
        return rng.integers(0, int(rng.choice([2, 3, 4])), size=n, dtype=np.uint8) + 48
    if k == 3:
        line = rng.integers(32, 127, size=int(rng.integers(20, 300)), dtype=np.uint8)
        a = np.tile(line, n // len(line) + 1)[:n].copy()
        if n: a[rng.integers(0, n, size=max(1, n // 5000))] ^= 0x20
        return a
    if k == 4:
        a = np.zeros(n, dtype=np.uint8)
        for _ in range(int(rng.integers(0, 12))):
            if n < 16: break
            s = int(rng.integers(0, n - 8)); ln = int(min(n - s, rng.integers(1, 3000)))
            a[s:s + ln] = rng.integers(0, 256, size=ln, dtype=np.uint8)
        return a
    if k == 5:
        return (np.arange(n // 4 + 1, dtype=np.uint32) * int(rng.choice([1, 3, 256, 65537]))).view(np.uint8)[:n].copy()
    if k == 6:
        h = shape(rng, n // 2)
        return np.concatenate([h, datagen(lo, n - len(h), int(rng.integers(5, 98)), int(rng.integers(0, 1 << 20)))])
    r = int(rng.choice([24, 96, 400]))
    a = np.repeat(rng.integers(0, 256, size=n // r + 1, dtype=np.uint8), r)[:n].copy()
    a[::int(rng.choice([97, 389, 1021]))] ^= 1
    return a


ctx = z.Context(max_units=256)
dctx = z.DContext()
bad = cases = units = frames = 0
slow = []
for t in range(trials):
    unit = int(rng.choice([131072, 131072, 65536, 20000, 4096, 100000]))
    n = int(rng.integers(max(1, unit // 3), 12 * unit))
    a = np.ascontiguousarray(shape(rng, n))
    level = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, -1, -5]))
    no_row = int(rng.integers(0, 2))
    ctx.set_row_matcher(2 if no_row else 0)
    t0 = time.time()
    try:
        comp, sizes = ctx.compress(a, level=level, unit_size=unit, return_sizes=True)
    except z.ZhipError:
        continue                                             # a strategy the device does not run (btlazy2 and above)
    dt = time.time() - t0
    if dt > 1.5: slow.append((round(dt, 2), t, level, unit, n))
    cases += 1
    if dctx.decompress(comp) != a.tobytes():
        bad += 1; print("DECODE MISMATCH trial", t, level, unit, n, flush=True)
    pos = 0
    for k in range(-(-n // unit)):
        u = a[k * unit: (k + 1) * unit]
        cp = (C.c_uint * 7)(); assert lo.zo_get_cparams(level, len(u), cp) == 0
        row = 3 <= cp[6] <= 5 and cp[0] > 14 and not no_row
        lo.zo_set_row_matcher(1 if row else 0)
        o = np.zeros(lo.zo_compress_bound(len(u)) + 64, dtype=np.uint8)
        rr = lo.zo_compress_unit(_buf(o), len(o), _buf(u), len(u), level)
        got = comp[pos: pos + int(sizes[k])]; pos += int(sizes[k])
        units += 1
        if rr == ERR or got != o[:rr].tobytes():
            bad += 1; print("UNIT MISMATCH trial", t, "unit", k, len(u), "level", level, "noRow", no_row, list(cp), flush=True)
            break
    lo.zo_set_row_matcher(0)
    if t % 2 == 0 and n >= 8:                                  # the same buffer as ONE multi-block frame
        cp = (C.c_uint * 7)(); assert lo.zo_get_cparams(level, n, cp) == 0
        if cp[6] <= 5:
            row = 3 <= cp[6] <= 5 and cp[0] > 14 and not no_row
            t0 = time.time()
            try:
                out = ctx.compress_frames([a], level)[0]
            except z.ZhipError:
                continue
            dt = time.time() - t0
            if dt > 1.5: slow.append((round(dt, 2), t, "frame", level, n))
            frames += 1
            if out != oracle_frame_params(lo, a, cp, row):
                bad += 1; print("FRAME MISMATCH trial", t, n, "level", level, "noRow", no_row, list(cp), flush=True)
            elif dctx.decompress(out) != a.tobytes():
                bad += 1; print("FRAME DECODE MISMATCH trial", t, n, level, flush=True)
print(f"seed {seed}: {cases} buffers, {units} units, {frames} frames, {bad} mismatches; calls above 1.5 s: {slow[:8]}", flush=True)
