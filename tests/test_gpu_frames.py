"""-m gpu: multi-block frames.  zhip_compress_frames and the shim's single-frame mode against the
committed digests of the REAL reference's single frame (tests/golden/frames_v1.json), the oracle's zo_compress_frame and —
when oracle/_ref travelled — the reference itself.  Done-criterion of the round: ZSTD_compress2 of 1 MiB through the shim,
byte-identical to the reference's frame at level 1."""
import ctypes as C
import hashlib
import json
import os
import numpy as np
import pytest
from _libs import (load_oracle, load_ref, have_ref, frame_cases, oracle_frame, mt_frame_cases, oracle_frame_mt, ref_frame_mt, MT_MODES,
                   datagen, text_like, _buf, ROOT, ERR)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "frames_v1.json")


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import zstd_amd
    zstd_amd.lib()
    return zstd_amd, load_oracle()


def test_frames_equal_the_reference_digests(env):
    z, lo = env
    gold = {(g["case"], g["level"]): g for g in json.load(open(GOLD))["frames"]}
    cases = list(frame_cases(lo))
    ctx = z.Context(max_units=64)
    seen = 0
    for level in (1, 2, 3, 4, -1, -5):
        todo = [(name, a) for name, a in cases if (name, level) in gold]
        outs = ctx.compress_frames([a for _, a in todo], level)          # one batch: the frames run side by side
        for (name, a), out in zip(todo, outs):
            g = gold[(name, level)]
            assert len(out) == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], (name, level)
            seen += 1
    assert seen == len(gold)


def test_frames_edge_sizes_and_checksum(env):
    z, lo = env
    ctx = z.Context(max_units=64)
    bufs = [np.zeros(0, np.uint8), datagen(lo, 3, 50, 1), datagen(lo, 131072, 50, 2), datagen(lo, 131072 + 7, 50, 3),
            datagen(lo, 2 * 131072, 30, 4), text_like(94208 + 131072, 5)]
    for level in (1, 3, -1):
        outs = ctx.compress_frames(bufs, level)
        for a, out in zip(bufs, outs):
            assert out == oracle_frame(lo, a, level), (len(a), level)
    ctx.set_checksum(True)
    a = datagen(lo, 500000, 50, 9)
    out = ctx.compress_frames([a], 1)[0]
    plain = oracle_frame(lo, a, 1)
    lo.zo_xxh64.restype = C.c_uint64
    lo.zo_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    assert out[4] == plain[4] | 4 and out[5:-4] == plain[5:]
    assert int.from_bytes(out[-4:], "little") == lo.zo_xxh64(_buf(a), a.size, 0) & 0xFFFFFFFF
    assert z.DContext().decompress(out) == a.tobytes()


def test_large_batch_of_frames_takes_the_hbm_table_form(env):
    """more workgroups than the LDS-table form holds at once (2 per CU) run on k_frame_hbm with every table in HBM (round 6): the same bytes as the
    oracle's frames, and as the same frames compressed a few at a time (the LDS form)"""
    z, lo = env
    import torch
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    nf = 2 * ncu + 40
    base = [datagen(lo, 131072 + 4096 * k, 40 + 5 * k, 70 + k) for k in range(4)] + [text_like(200000, 3), datagen(lo, 300, 50, 4), np.zeros(0, np.uint8)]
    bufs = [base[i % len(base)] for i in range(nf)]
    big = z.Context(max_units=nf)
    outs = big.compress_frames(bufs, 1)
    want = [oracle_frame(lo, a, 1) for a in base]
    for i, out in enumerate(outs):
        assert out == want[i % len(base)], i
    few = z.Context(max_units=16).compress_frames(base, 1)                # 7 workgroups: tables in LDS
    assert few == want


def test_frames_decode_on_the_device_and_unsupported_strategy(env):
    z, lo = env
    ctx = z.Context(max_units=8)
    a = datagen(lo, 1 << 20, 50, 21)
    out = ctx.compress_frames([a], 1)[0]
    assert z.DContext().decompress(out) == a.tobytes()                              # the device decoder reads multi-block frames too
    out3 = ctx.compress_frames([a], 3)[0]                                # level 3 (ZSTD_dfast, the default level) too
    assert out3 == oracle_frame(lo, a, 3) and z.DContext().decompress(out3) == a.tobytes()
    with pytest.raises(z.ZhipError):
        ctx.compress_frames([a], 13)                                     # btlazy2: not a device strategy, and there is no CPU fallback (levels 5..12 are tests/test_gpu_frames_lazy.py's)


def test_shim_single_frame_mode_1mib_level1(env):
    """ZSTD_compress2 of 1 MiB through the drop-in with ZHIP_c_singleFrame: the reference's single frame, byte for byte"""
    z, lo = env
    from zstd_amd import build as zb
    S = C.CDLL(zb.SHIM)
    S.ZSTD_createCCtx.restype = C.c_void_p
    S.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    S.ZSTD_CCtx_setParameter.restype = C.c_size_t
    S.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    S.ZSTD_compress2.restype = C.c_size_t
    S.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    S.ZSTD_compressBound.restype = C.c_size_t
    S.ZSTD_compressBound.argtypes = [C.c_size_t]
    S.ZSTD_isError.argtypes = [C.c_size_t]
    lr = load_ref() if have_ref() else None
    gold = {(g["case"], g["level"]): g for g in json.load(open(GOLD))["frames"]}
    for name, a in (("dg_1m", datagen(lo, 1 << 20, 50, 6)), ("text_700k", text_like(700000, 3))):
        c = S.ZSTD_createCCtx()
        assert S.ZSTD_CCtx_setParameter(c, 100, 1) == 0
        assert S.ZSTD_CCtx_setParameter(c, 100001, 1) == 0               # ZHIP_c_singleFrame
        cap = S.ZSTD_compressBound(a.size)
        dst = np.zeros(cap, dtype=np.uint8)
        r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), a.size)
        assert not S.ZSTD_isError(r)
        out = dst[:r].tobytes()
        g = gold[(name, 1)]
        assert r == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], name
        assert out == oracle_frame(lo, a, 1)
        if lr is not None:
            lr.zref_compress_frame.restype = C.c_size_t
            lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            want = np.zeros(cap + 1024, dtype=np.uint8)
            k = lr.zref_compress_frame(1, _buf(a), a.size, _buf(want), len(want))
            assert out == want[:k].tobytes()
        # level 3 (ZSTD_dfast, the default level): the reference's single frame too
        assert S.ZSTD_CCtx_setParameter(c, 100, 3) == 0
        r3 = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), a.size)
        assert not S.ZSTD_isError(r3) and dst[:r3].tobytes() == oracle_frame(lo, a, 3)
        # level 5 (greedy) keeps the frame-per-unit stream: still a valid zstd stream of the same content
        assert S.ZSTD_CCtx_setParameter(c, 100, 5) == 0
        r5 = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), a.size)
        assert not S.ZSTD_isError(r5) and z.DContext().decompress(dst[:r5].tobytes()) == a.tobytes()
        S.ZSTD_freeCCtx(c)


def test_frames_fuzz_against_oracle(env):
    """random sizes / contents / levels in one batch per level: block splits at 128 KB / 92 KB, raw and RLE blocks in any order,
    matches across block borders and beyond the window (level 1 above 512 KB), ragged last blocks"""
    z, lo = env
    rng = np.random.default_rng(20260923)
    ctx = z.Context(max_units=64)

    def piece(kind, n, seed):
        if kind == 0:
            return datagen(lo, n, int(rng.integers(5, 98)), seed)
        if kind == 1:
            return text_like(n, seed)
        if kind == 2:
            return rng.integers(0, 256, size=n, dtype=np.uint8)
        if kind == 3:
            return np.full(n, int(rng.integers(0, 256)), dtype=np.uint8)
        return rng.integers(0, 3, size=n, dtype=np.uint8)

    bufs = []
    for t in range(40):
        n = int(rng.integers(1, 1_600_000))
        parts, left = [], n
        while left > 0:
            k = int(min(left, rng.integers(1, 400_000)))
            parts.append(piece(int(rng.integers(0, 5)), k, t * 17 + len(parts)))
            left -= k
        a = np.concatenate(parts)
        if t % 5 == 0 and n > 300_000:                       # a far repeat: same content 600 KB later (outside level 1's window of 512 KB when n allows)
            m = min(n // 3, 200_000)
            a[n - m:] = a[:m]
        bufs.append(a)
    for level in (1, 3, -2):
        outs = ctx.compress_frames(bufs, level)
        for i, (a, out) in enumerate(zip(bufs, outs)):
            cp = (C.c_uint * 7)()
            assert lo.zo_get_cparams(level, len(a), cp) == 0
            if cp[6] not in (1, 2):
                continue
            assert out == oracle_frame(lo, a, level), (i, len(a), level)


def test_frames_with_explicit_parameters(env):
    """explicit windowLog / hashLog on a multi-block frame: a window smaller than the input (matches beyond it are refused, the
    frame header carries a window descriptor), a table above the LDS bound (hashLog 15 -> HBM), a wider hash — against the
    oracle's frame loop run with the same effective parameters (and the reference itself where oracle/_ref travelled)"""
    z, lo = env
    L = z.lib()
    L.zhip_getCParams_explicit.restype = C.c_int
    L.zhip_getCParams_explicit.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p]
    lo.zo_compress_frame_params.restype = C.c_size_t
    lo.zo_compress_frame_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_frame_bound.restype = C.c_size_t
    lo.zo_frame_bound.argtypes = [C.c_size_t]
    ctx = z.Context(max_units=8)
    a = datagen(lo, 900_000, 70, 33).copy()
    a[600_000:700_000] = a[100_000:200_000]                    # a repeat 500 000 bytes back: inside a 2^19 window, outside a 2^17 one
    for level, cp in ((1, [17, 0, 0, 0, 0, 0, 0]), (1, [18, 0, 15, 0, 4, 0, 0]), (3, [17, 15, 16, 0, 0, 0, 0]), (1, [0, 0, 13, 0, 6, 2, 0])):
        eff = (C.c_uint * 7)()
        req = (C.c_uint * 7)(*cp)
        assert L.zhip_getCParams_explicit(level, a.size, req, eff) == 0
        out = ctx.compress_frames([a], level, cparams=cp)[0]
        cap = lo.zo_frame_bound(a.size)
        want = np.zeros(cap, dtype=np.uint8)
        r = lo.zo_compress_frame_params(_buf(want), cap, _buf(a), a.size, eff)
        assert r != ERR and out == want[:r].tobytes(), (level, cp, list(eff))
        assert z.DContext().decompress(out) == a.tobytes()


GOLD_MT = os.path.join(os.path.dirname(__file__), "golden", "frames_mt_v1.json")


def test_job_pool_frames_equal_the_reference_digests(env):
    """zhip_compress_frames_mt = ZSTD_compress2 with ZSTD_c_nbWorkers >= 1: every case of frames_mt_v1.json (digests of the REAL
    reference), one batch per mode — jobs of all the frames run side by side"""
    z, lo = env
    gold = {(g["case"], g["level"], g["jobSize"], g["overlapLog"], g["checksum"]): g for g in json.load(open(GOLD_MT))["frames"]}
    cases = list(mt_frame_cases(lo))
    ctx = z.Context(max_units=64)
    seen = 0
    for level, js, ov, ck in MT_MODES:
        ctx.set_checksum(bool(ck))
        outs = ctx.compress_frames([a for _, a in cases], level, workers=1, job_size=js, overlap_log=ov)
        for (name, a), out in zip(cases, outs):
            g = gold[(name, level, js, ov, ck)]
            assert len(out) == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], (name, level, js, ov, ck)
            seen += 1
    assert seen == len(gold)
    ctx.set_checksum(False)
    a = cases[1][1]
    out = ctx.compress_frames([a], 1, workers=4)[0]                      # the worker count does not change the bytes
    assert out == oracle_frame_mt(lo, a, 1) and z.DContext().decompress(out) == a.tobytes()


def test_job_pool_frames_mixed_batch_and_small_inputs(env):
    """inputs at or below 512 KB come out as the plain single-context frame (the reference drops the workers there); large and
    small inputs share a batch; frame sizes are per input, not per job"""
    z, lo = env
    ctx = z.Context(max_units=64)
    bufs = [np.zeros(0, np.uint8), datagen(lo, 100, 50, 1), datagen(lo, 524288, 50, 2), datagen(lo, 524289, 50, 3), text_like(1_500_000, 4),
            datagen(lo, 300_000, 50, 5), datagen(lo, 2_500_000, 30, 6)]
    for level in (1, 3):
        outs = ctx.compress_frames(bufs, level, workers=2, job_size=524288)
        for a, out in zip(bufs, outs):
            want = oracle_frame(lo, a, level) if len(a) <= 524288 else oracle_frame_mt(lo, a, level, 524288, 0)
            assert out == want, (len(a), level)
    assert z.DContext().decompress(b"".join(outs)) == b"".join(a.tobytes() for a in bufs)
    with pytest.raises(z.ZhipError):
        ctx.compress_frames([bufs[-1]], 13, workers=1)                    # btlazy2: not a device strategy, no CPU fallback
    small = z.Context(max_units=2)
    with pytest.raises(z.ZhipError):
        small.compress_frames([bufs[-1]], 1, workers=1, job_size=524288)  # 5 jobs > the context's capacity


def test_shim_nbworkers_mode(env):
    """ZSTD_compress2 through the drop-in with ZSTD_c_nbWorkers / ZSTD_c_jobSize / ZSTD_c_overlapLog / ZSTD_c_checksumFlag: the
    reference's multi-threaded frame, byte for byte"""
    z, lo = env
    from zstd_amd import build as zb
    S = C.CDLL(zb.SHIM)
    S.ZSTD_createCCtx.restype = C.c_void_p
    S.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    S.ZSTD_CCtx_setParameter.restype = C.c_size_t
    S.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    S.ZSTD_compress2.restype = C.c_size_t
    S.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    S.ZSTD_compressCCtx.restype = C.c_size_t
    S.ZSTD_compressCCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    S.ZSTD_compressBound.restype = C.c_size_t
    S.ZSTD_compressBound.argtypes = [C.c_size_t]
    S.ZSTD_isError.argtypes = [C.c_size_t]
    lr = load_ref() if have_ref() else None
    a = datagen(lo, 3_000_000, 50, 41)
    cap = S.ZSTD_compressBound(a.size)
    dst = np.zeros(cap, dtype=np.uint8)
    c = S.ZSTD_createCCtx()
    assert S.ZSTD_isError(S.ZSTD_CCtx_setParameter(c, 400, 257)) and S.ZSTD_isError(S.ZSTD_CCtx_setParameter(c, 402, 10))   # out of bounds
    for level, workers, js, ov, ck in ((1, 1, 0, 0, 0), (3, 8, 1 << 20, 4, 1), (-1, 2, 524288, 9, 0)):
        for prm, v in ((100, level), (400, workers), (401, js), (402, ov), (201, ck)):
            assert S.ZSTD_CCtx_setParameter(c, prm, v) == 0
        r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), a.size)
        assert not S.ZSTD_isError(r)
        out = dst[:r].tobytes()
        assert out == oracle_frame_mt(lo, a, level, js, ov, bool(ck)), (level, js, ov, ck)
        if lr is not None:
            assert out == ref_frame_mt(lr, a, level, js, ov, bool(ck)), (level, js, ov, ck)
    # at or below 512 KB: the workers are dropped -> the frame-per-unit stream of the default mode; ZSTD_compressCCtx ignores the workers
    b = a[:400_000]
    r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(b), b.size)
    assert not S.ZSTD_isError(r) and z.DContext().decompress(dst[:r].tobytes()) == b.tobytes()
    r = S.ZSTD_compressCCtx(c, _buf(dst), cap, _buf(a), a.size, 1)
    assert not S.ZSTD_isError(r) and dst[:r].tobytes() != oracle_frame_mt(lo, a, 1) and z.DContext().decompress(dst[:r].tobytes()) == a.tobytes()
    S.ZSTD_freeCCtx(c)


def test_job_window_starting_at_the_frames_first_byte(env):
    """jobSize <= overlap: the second job's window starts at the frame's byte 0, which the reference can match from that job; the frame
    is the FIRST thing in its device allocation here, so a stray read in front of it would fault"""
    z, lo = env
    ctx = z.Context(max_units=16)
    for level, js, ov in ((2, 524288, 8), (3, 524288, 7), (1, 524288, 9)):
        for seed in (10, 11):
            a = datagen(lo, 700000, 20, seed).copy()
            a[js: js + 64] = a[0: 64]
            out = ctx.compress_frames([a], level, workers=1, job_size=js, overlap_log=ov)[0]
            assert out == oracle_frame_mt(lo, a, level, js, ov), (level, seed)


def test_job_pool_frames_with_explicit_parameters(env):
    """ZSTD_c_nbWorkers + explicit compression parameters (windows smaller than a job, table logs 12-17 = LDS 24-bit, LDS / HBM 32-bit
    tables, both strategies) against the oracle, which tests/test_oracle_vs_reference.py pins to the reference on the same cases"""
    z, lo = env
    from test_oracle_vs_reference import mt_explicit_cases
    ctx = z.Context(max_units=16)
    seen = 0
    for a, level, req, eff, js, ov, ck in mt_explicit_cases(lo, 40, 7):
        ctx.set_checksum(ck)
        out = ctx.compress_frames([a], level, cparams=req, workers=1, job_size=js, overlap_log=ov)[0]
        assert out == oracle_frame_mt(lo, a, level, js, ov, ck, cp=eff), (len(a), level, req, js, ov, ck)
        seen += 1
    ctx.set_checksum(False)
    assert seen >= 24
