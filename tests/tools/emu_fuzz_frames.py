#!/usr/bin/env python3
"""tests/tools/emu_fuzz_frames.py <jobs|frames|lazyframes|lazyjobs> <seed> <seconds> — no GPU: the frame kernel on the host SIMT emulator
(tests/simt/libzhip_emu.so, the product's device code compiled for the CPU) against the oracle, on random inputs x random effective
parameters (strategies fast / dfast, windowLog 17..23: below 17 the host refuses, `zhip_lib.hip: "windowLog below the block size"`, and
so does the oracle).  `jobs`: one input above 512 KB as a job-pool frame (random job size, overlap, checksum); `frames`: batches of 1-3
multi-block frames (1 B .. 900 KB).  Prints BAD lines and saves the input under /tmp; `done <seed> <cases>` at the end.
`lazyframes` / `lazyjobs`: the same for the strategies greedy / lazy / lazy2 (zhip_frame_lazy.h), row matcher or hash chain, with and without the
two-pass prediction.  The oracle is pinned to the reference on the same parameter domain by tests/test_oracle_vs_reference.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
from _libs import (load_oracle, load_emu, datagen, text_like, oracle_frame_mt, emu_compress_frame_jobs, emu_compress_frames,
                   oracle_frame_params, emu_compress_frames_lazy, emu_compress_frame_jobs_lazy, _buf)

mode, seed, tmax = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
lo, le = load_oracle(), load_emu()
lo.zo_xxh64.restype = C.c_uint64
lo.zo_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
rng = np.random.default_rng(seed)


def mk(n, kind, s, piece=400000):
    if kind == 0:
        return datagen(lo, n, int(rng.choice([20, 50, 80, 95])), s)
    if kind == 1:
        return text_like(n, s)
    if kind == 2:
        return rng.integers(0, 256, size=n, dtype=np.uint8)
    if kind == 3:                                                       # a third of the input repeats the first third: far matches, overlaps
        a = datagen(lo, n, 50, s).copy(); k = n // 3; a[k:2 * k] = a[:k]; return a
    if kind == 4:                                                       # zeros with a few ones: RLE blocks, huge matches
        a = np.zeros(n, np.uint8); a[rng.integers(0, n, size=max(1, n // 5000))] = 1; return a
    parts, left = [], n
    while left > 0:
        m = min(left, int(rng.integers(100, piece))); parts.append(mk(m, int(rng.integers(0, 5)), s + len(parts))); left -= m
    return np.concatenate(parts)


def params():
    wl = int(rng.integers(17, 24))
    return [wl, int(rng.integers(6, wl + 1)), int(rng.integers(6, min(wl + 1, 19) + 1)), 1, int(rng.integers(3, 8)),
            int(rng.choice([0, 1, 2, 8, 16])), int(rng.choice([1, 2]))]


def lazy_params():
    wl = int(rng.integers(17, 22))
    hl = int(rng.integers(10, min(wl + 1, 20) + 1))
    return [wl, int(rng.integers(8, wl + 1)), hl, int(rng.integers(1, 7)), int(rng.integers(3, 8)), int(rng.choice([0, 2, 8, 16])), int(rng.choice([3, 4, 5]))]


t0, cases, bad = time.time(), 0, 0
while time.time() - t0 < tmax:
    ck = bool(rng.integers(0, 2))
    if mode in ("lazyframes", "lazyjobs"):
        cpl = lazy_params(); row = int(rng.integers(0, 2)); pred = int(rng.integers(0, 2))
        os.environ["ZHIP_LZ_PREDICT"] = str(pred)
        lo.zo_set_row_matcher.argtypes = [C.c_int]
        if mode == "lazyframes":
            bufs = [np.ascontiguousarray(mk(int(rng.choice([rng.integers(1, 4000), rng.integers(100000, 140000), rng.integers(131073, 500000)])),
                                            int(rng.integers(0, 6)), int(rng.integers(0, 1 << 30)), 150000)) for _ in range(int(rng.integers(1, 4)))]
            for b, g in zip(bufs, emu_compress_frames_lazy(le, lo, bufs, [cpl] * len(bufs), row)):
                want = oracle_frame_params(lo, b, (C.c_uint * 7)(*cpl), row)
                if g != want:
                    bad += 1; print("BAD", mode, seed, len(b), cpl, row, pred, flush=True)
                    np.save(f"/tmp/emu_fuzz_bad_{seed}_{cases}.npy", b)
        else:
            a = np.ascontiguousarray(mk(int(rng.integers(524289, 1_300_000)), int(rng.integers(0, 6)), int(rng.integers(0, 1 << 30)), 250000))
            js = int(rng.choice([0, 524288, 524288, int(rng.integers(524288, 900000))])); ov = int(rng.choice([0, 0, 1, 3, 6, 8, 9]))
            lo.zo_set_row_matcher(row)
            try:
                want = oracle_frame_mt(lo, a, 0, js, ov, ck, cp=(C.c_uint * 7)(*cpl))
            finally:
                lo.zo_set_row_matcher(0)
            got = emu_compress_frame_jobs_lazy(le, lo, a, (C.c_uint * 7)(*cpl), row, js, ov, ck)
            if bytes(got) != bytes(want):
                bad += 1; print("BAD", mode, seed, len(a), cpl, row, pred, js, ov, ck, flush=True)
                np.save(f"/tmp/emu_fuzz_bad_{seed}_{cases}.npy", a)
    elif mode == "jobs":
        a = np.ascontiguousarray(mk(int(rng.integers(524289, 2_600_000)), int(rng.integers(0, 6)), int(rng.integers(0, 1 << 30))))
        level = int(rng.choice([1, 1, 2, 3, 3, -1, -5, 4]))
        js = int(rng.choice([0, 524288, 524288, int(rng.integers(524288, 1500000))]))
        ov = int(rng.choice([0, 0, 1, 3, 6, 8, 9]))
        cp = (C.c_uint * 7)(*params()) if rng.integers(0, 3) == 0 else None
        want = oracle_frame_mt(lo, a, level, js, ov, ck, cp=cp)
        got = emu_compress_frame_jobs(le, lo, a, level, js, ov, ck, cp=cp)
        if bytes(got) != bytes(want):
            bad += 1; print("BAD", seed, len(a), level, js, ov, ck, list(cp) if cp else None, flush=True)
            np.save(f"/tmp/emu_fuzz_bad_{seed}_{cases}.npy", a)
    else:
        bufs = [np.ascontiguousarray(mk(int(rng.choice([rng.integers(1, 4000), rng.integers(100000, 140000), rng.integers(131073, 900000)])),
                                        int(rng.integers(0, 6)), int(rng.integers(0, 1 << 30)), 200000)) for _ in range(int(rng.integers(1, 4)))]
        cpl = params()
        for b, g in zip(bufs, emu_compress_frames(le, lo, bufs, 1, ck, cparams=cpl)):
            want = oracle_frame_params(lo, b, (C.c_uint * 7)(*cpl), False)
            if ck:                                                      # the checksum flag of the frame header + XXH64's low 32 bits
                want = want[:4] + bytes([want[4] | 4]) + want[5:] + int(lo.zo_xxh64(_buf(b), len(b), 0) & 0xffffffff).to_bytes(4, "little")
            if g != want:
                bad += 1; print("BAD", seed, len(b), cpl, ck, flush=True)
                np.save(f"/tmp/emu_fuzz_bad_{seed}_{cases}.npy", b)
    cases += 1
print("done", seed, cases, "bad", bad, flush=True)
sys.exit(1 if bad else 0)
