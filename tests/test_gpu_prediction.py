"""-m gpu: the row matcher's two-pass prediction switched on through zhip_set_prediction (off by default): units at the lazy levels against the
committed reference digests and the oracle, multi-block / job-pool frames against the reference digests — same bytes as without it.
The unit form passed tests/test_gpu_rowhash.py's compressing tests on MI355X with the prediction on; this file sorts last because the frame
form (k_lz_predict) has not run on a GPU yet (the round's GPU budget ended first)."""
import ctypes as C
import hashlib
import json
import os
import numpy as np
import pytest
from _libs import (load_oracle, corpus_cases, lazy_frame_cases, LAZY_FRAME_MODES, oracle_frame_mt, datagen, text_like, _buf, ROOT, ERR)

pytestmark = pytest.mark.gpu
UNIT = 131072


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import zstd_amd
    zstd_amd.lib()
    return zstd_amd, load_oracle()


def test_units_with_prediction_equal_the_reference_digests(env):
    z, lo = env
    ctx = z.Context(0, max_units=64)
    ctx.set_row_matcher(0)
    ctx.set_prediction(units=1)
    gold = {(g["case"], g["level"]): g for g in json.load(open(os.path.join(ROOT, "tests", "golden", "units_v3_rowhash.json")))["units"]}
    seen = 0
    for n in (131072, 40000):
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(0,)):
            for level in (5, 7, 8, 10):
                g = gold.get((name, level))
                if g is None:
                    continue
                got = ctx.compress(a, level=level)
                assert len(got) == g["csize"] and hashlib.sha256(got).hexdigest() == g["dst_sha256"], (name, level)
                seen += 1
    assert seen > 0
    # a batch with long matches, lazy skipping and a 5-byte unit: against the oracle, and the same bytes as with the prediction off
    rng = np.random.default_rng(1)
    a = np.concatenate([datagen(lo, 6 * UNIT, 35, 6), np.tile(rng.integers(0, 256, 900, dtype=np.uint8), 300)[: 2 * UNIT], text_like(2 * UNIT, 4), datagen(lo, UNIT + 5, 50, 2)])
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    lo.zo_set_row_matcher(1)
    try:
        for level in (5, 8):
            got = ctx.compress(a, level=level)
            cap = lo.zo_compress_bound(UNIT) * 16
            want = np.empty(cap, dtype=np.uint8)
            r = lo.zo_compress_chunks(level, UNIT, _buf(a), len(a), _buf(want), cap, None, 0)
            assert r != ERR and got == want[:r].tobytes(), level
            ctx.set_prediction(units=0)
            assert ctx.compress(a, level=level) == got
            ctx.set_prediction(units=1)
    finally:
        lo.zo_set_row_matcher(0)


def test_frames_with_prediction_equal_the_reference_digests(env):
    z, lo = env
    gold = {(g["case"], g["level"], g["noRow"]): g for g in json.load(open(os.path.join(ROOT, "tests", "golden", "frames_lazy_v1.json")))["frames"]}
    cases = [(name, a) for name, a in lazy_frame_cases(lo) if name in ("dg_400000", "text_700k", "mixed_900k", "zeros_500k", "tail_10")]
    ctx = z.Context(max_units=64)
    ctx.set_prediction(frames=1)
    for level, no_row in ((5, 0), (6, 0), (7, 1), (8, 0), (10, 0)):
        ctx.set_row_matcher(2 if no_row else 0)
        outs = ctx.compress_frames([a for _, a in cases], level)
        for (name, a), out in zip(cases, outs):
            g = gold[(name, level, no_row)]
            assert len(out) == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], (name, level, no_row)
    # a job-pool frame (every job predicts its own window)
    ctx.set_row_matcher(0)
    a = np.concatenate([datagen(lo, 2 << 20, 50, 15), text_like(1 << 20, 13)])
    out = ctx.compress_frames([a], 5, workers=2, job_size=1 << 20)[0]
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    lo.zo_set_row_matcher(1)
    try:
        assert out == oracle_frame_mt(lo, a, 5, 1 << 20, 0, 0)
    finally:
        lo.zo_set_row_matcher(0)
