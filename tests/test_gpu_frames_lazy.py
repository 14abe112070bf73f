"""-m gpu: multi-block frames and job-pool frames of the strategies greedy / lazy / lazy2 (levels 5-10; SURVEY.md §8f rank 1, the
round-2 verdict's item 4).  zhip_compress_frames / zhip_compress_frames_mt and the shim's ZSTD_compress2 against the committed digests
of the REAL reference's frames (tests/golden/frames_lazy_v1.json: 49 frames, the row matcher and the hash chain), the oracle, and —
when oracle/_ref travelled — the reference itself with ZSTD_c_nbWorkers."""
import ctypes as C
import hashlib
import json
import os
import numpy as np
import pytest
from _libs import (load_oracle, load_ref, have_ref, lazy_frame_cases, LAZY_FRAME_MODES, oracle_frame_params, oracle_frame_mt, ref_frame_mt,
                   datagen, text_like, _buf, ROOT, ERR)

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "frames_lazy_v1.json")


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import zstd_amd
    zstd_amd.lib()
    return zstd_amd, load_oracle()


def test_lazy_frames_equal_the_reference_digests(env):
    z, lo = env
    gold = {(g["case"], g["level"], g["noRow"]): g for g in json.load(open(GOLD))["frames"]}
    cases = list(lazy_frame_cases(lo))
    ctx = z.Context(max_units=64)
    seen = 0
    for level, no_row in LAZY_FRAME_MODES:
        ctx.set_row_matcher(2 if no_row else 0)
        outs = ctx.compress_frames([a for _, a in cases], level)          # one batch: the frames run side by side
        for (name, a), out in zip(cases, outs):
            g = gold[(name, level, no_row)]
            assert len(out) == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], (name, level, no_row)
            seen += 1
    assert seen == len(gold)
    d = z.DContext()
    assert d.decompress(outs[0]) == cases[0][1].tobytes()


def test_lazy_frames_explicit_parameters_and_small_windows(env):
    """inputs larger than the window, both matchers, against the oracle (pinned to the reference by test_oracle_vs_reference.py)"""
    z, lo = env
    ctx = z.Context(max_units=64)
    rng = np.random.default_rng(3)
    bufs = [datagen(lo, 900000, 70, 3), np.concatenate([datagen(lo, 140000, 50, 2)] * 5), text_like(500000, 4),
            np.concatenate([rng.integers(0, 256, size=200000, dtype=np.uint8), np.zeros(300000, np.uint8), datagen(lo, 200000, 30, 8)])]
    for cp, row in (([17, 16, 17, 3, 5, 2, 3], 1), ([17, 15, 16, 4, 4, 8, 4], 0), ([17, 17, 18, 4, 5, 16, 5], 1), ([18, 17, 18, 5, 4, 32, 5], 0),
                    ([19, 12, 13, 2, 6, 4, 4], 1)):
        ctx.set_row_matcher(0 if row else 2)
        outs = ctx.compress_frames(bufs, 5, cparams=cp)
        for a, out in zip(bufs, outs):
            assert out == oracle_frame_params(lo, a, (C.c_uint * 7)(*cp), row), (cp, row, len(a))


def test_frames_of_short_runs_at_greedy(env):
    """every byte 24 times: each sequence is a repcode that greedy takes without a search, so nextToUpdate never moves — the 384-position rule used to flag the whole
    gap behind it again at every batch start (9 s per MiB of such a frame until round 6).  The oracle's frame, and a time bound far above what it takes now"""
    import time
    z, lo = env
    ctx = z.Context(max_units=8)
    rng = np.random.default_rng(24)
    a = np.repeat(rng.integers(0, 256, size=(1 << 20) // 24 + 1, dtype=np.uint8), 24)[: 1 << 20]
    b = np.repeat(rng.integers(0, 256, size=300000 // 1000 + 1, dtype=np.uint8), 1000)[:300000]
    ctx.compress_frames([a[:70000]], 5)                                  # (the first call of a process loads the code object and sizes the arenas)
    for level, row in ((5, 1), (6, 1), (5, 0)):
        ctx.set_row_matcher(0 if row else 2)                             # (the suite's default is the hash chain: tests/conftest.py)
        cp = (C.c_uint * 7)(); assert lo.zo_get_cparams(level, len(a), cp) == 0
        t0 = time.time()
        outs = ctx.compress_frames([a, b], level)
        dt = time.time() - t0
        assert outs[0] == oracle_frame_params(lo, a, cp, row), (level, row)
        cpb = (C.c_uint * 7)(); assert lo.zo_get_cparams(level, len(b), cpb) == 0
        assert outs[1] == oracle_frame_params(lo, b, cpb, row), (level, row)
        assert dt < 3.0, (level, row, dt)


@pytest.mark.parametrize("level,no_row,js,ov,ck", [(5, 0, 0, 0, 0), (6, 1, 524288, 9, 1), (8, 0, 700000, 0, 0), (10, 1, 1 << 20, 3, 1)])
def test_lazy_job_pool_frames(env, level, no_row, js, ov, ck):
    """ZSTD_c_nbWorkers >= 1 at the lazy levels: a workgroup per job; against the oracle and, when it travelled, the reference"""
    z, lo = env
    ctx = z.Context(max_units=64)
    ctx.set_row_matcher(2 if no_row else 0)
    ctx.set_checksum(bool(ck))
    rng = np.random.default_rng(9)
    bufs = [datagen(lo, 5 << 20, 50, 15), text_like(3_000_000, 13),
            np.concatenate([datagen(lo, 700000, 50, 1), rng.integers(0, 256, size=500000, dtype=np.uint8), np.full(600000, 7, np.uint8), text_like(800000, 9)])]
    outs = ctx.compress_frames(bufs, level, workers=4, job_size=js, overlap_log=ov)
    lr = load_ref() if have_ref() else None
    for a, out in zip(bufs, outs):
        lo.zo_set_row_matcher(0 if no_row else 1)
        try:
            want = oracle_frame_mt(lo, a, level, js, ov, ck)
        finally:
            lo.zo_set_row_matcher(0)
        assert out == want, (level, no_row, js, ov, ck, len(a))
        if lr is not None and not no_row:
            assert out == ref_frame_mt(lr, a, level, js, ov, ck), ("reference", level, js, ov, ck, len(a))
    assert z.DContext().decompress(outs[1]) == bufs[1].tobytes()


def test_live_rows_switch_gives_the_same_bytes(env):
    """the row matcher's live rows (LzRing, on by default) against the walk through the links they replace (zhip_set_live_rows(0), also what a context
    falls back to when the rows' arena cannot be allocated): same bytes, and the oracle's, on long-match data — frames and a job-pool frame"""
    z, lo = env
    bufs = [datagen(lo, 1 << 20, 50, 5), datagen(lo, 700000, 80, 6), text_like(400000, 3)]
    whole = np.concatenate(bufs)
    outs = {}
    for ring in ("1", "0"):
        ctx = z.Context(max_units=64)
        ctx.set_live_rows(ring == "1")
        ctx.set_row_matcher(0)
        outs[ring] = (ctx.compress_frames(bufs, 5), ctx.compress_frames([whole], 7, workers=2, job_size=1 << 20), ctx.compress_frames(bufs[:2], 5, cparams=[20, 16, 17, 6, 5, 2, 5]))
        ctx.close()
    assert outs["1"] == outs["0"]
    lo.zo_set_row_matcher(1)
    try:
        assert outs["1"][1][0] == oracle_frame_mt(lo, whole, 7, 1 << 20, 0, 0)
        cp = (C.c_uint * 7)(); assert lo.zo_get_cparams(5, len(bufs[0]), cp) == 0
    finally:
        lo.zo_set_row_matcher(0)
    assert outs["1"][0][0] == oracle_frame_params(lo, bufs[0], cp, 1)
    assert outs["1"][2][1] == oracle_frame_params(lo, bufs[1], (C.c_uint * 7)(20, 16, 17, 6, 5, 2, 5), 1)          # rowLog 6: 63 entries per row
