"""Multi-block frames and job-pool frames of the strategies greedy / lazy / lazy2 on the host SIMT emulator: k_lz_links + k_lz_search +
k_frame_lazy (zstd_amd/csrc/zhip_frame_lazy.h) against the oracle's zo_compress_frame_params / zo_compress_frame_mt_params, which are
pinned to the reference (tests/golden/frames_lazy_v1.json, test_oracle_vs_reference.py).  Both match finders: the row hash (the
reference's default above windowLog 14) and the hash chain."""
import numpy as np
import pytest
from _libs import *
from _libs import _buf


@pytest.fixture(scope="module")
def libs():
    return load_oracle(), load_emu()


def _cp(lo, level, n):
    cp = (C.c_uint * 7)()
    assert lo.zo_get_cparams(level, n, cp) == 0
    return list(cp)


def _cases(lo):
    rng = np.random.default_rng(55)
    return [("dg_150k", datagen(lo, 150000, 50, 5)),
            ("mixed_310k", np.concatenate([datagen(lo, 70000, 50, 1), rng.integers(0, 256, size=90000, dtype=np.uint8),      # lazy skipping, raw blocks,
                                           np.full(80000, 7, np.uint8), text_like(70000, 9)])),                                 # long matches (the 384 / 192 rule), splitter borders
            ("tail_10", datagen(lo, 131072 + 10, 50, 9))]                                                                       # last block below the row matcher's guard


@pytest.mark.parametrize("level,row", [(5, 1), (8, 0)])
def test_lazy_frames_match_oracle(libs, level, row):
    lo, le = libs
    cases = _cases(lo)
    bufs = [a for _, a in cases]
    cps = [_cp(lo, level, len(a)) for a in bufs]
    got = emu_compress_frames_lazy(le, lo, bufs, cps, row)
    for (name, a), cp, g in zip(cases, cps, got):
        assert cp[6] in (3, 4, 5)
        assert g == oracle_frame_params(lo, a, (C.c_uint * 7)(*cp), row), (name, level, row)


def test_greedy_frame_of_runs_across_block_borders(libs):
    """runs of 24 with a byte flipped every 1 021: greedy takes repcode after repcode without a search, nextToUpdate lags, the batch starts flag the gap behind it
    (384-position rule) for a search that never comes — and at the next BLOCK start the reference moves nextToUpdate up (zstd_compress.c:3243) and later inserts what lies
    behind it.  The device used to keep those positions flagged (a parity bug older than round 6; found by tests/tools/gpu_fuzz_shapes.py seed 7002)."""
    lo, le = libs
    rng = np.random.default_rng(7002)
    n = 400000
    a = np.repeat(rng.integers(0, 256, size=n // 24 + 1, dtype=np.uint8), 24)[:n].copy()
    a[::1021] ^= 1
    b = np.repeat(rng.integers(0, 256, size=300000 // 96 + 1, dtype=np.uint8), 96)[:300000].copy()
    b[::389] ^= 1
    for row in (1, 0):
        cps = [_cp(lo, 5, len(x)) for x in (a, b)]
        got = emu_compress_frames_lazy(le, lo, [a, b], cps, row)
        for x, cp, g in zip((a, b), cps, got):
            assert g == oracle_frame_params(lo, x, (C.c_uint * 7)(*cp), row), (len(x), row)


@pytest.mark.parametrize("cp,row", [([17, 16, 17, 3, 5, 2, 4], 1), ([17, 17, 18, 4, 4, 16, 5], 0)])
def test_lazy_frames_beyond_the_window(libs, cp, row):
    """inputs larger than 2^windowLog: the window's low end moves with the blocks (candidates, repcodes, catch-up)"""
    lo, le = libs
    bufs = [datagen(lo, 200000, 70, 3)] if row else [datagen(lo, 300000, 70, 3), np.concatenate([datagen(lo, 140000, 50, 2)] * 2)]
    got = emu_compress_frames_lazy(le, lo, bufs, [cp] * len(bufs), row)
    for a, g in zip(bufs, got):
        assert g == oracle_frame_params(lo, a, (C.c_uint * 7)(*cp), row), (cp, row, len(a))


@pytest.mark.parametrize("level,row,js,ov,ck", [(6, 0, 524288, 9, 1)])
def test_lazy_job_pool_frames_match_oracle(libs, level, row, js, ov, ck):
    """ZSTD_c_nbWorkers semantics: every job indexes its whole prefix (all positions but the last 8), starts with zero repcodes"""
    lo, le = libs
    rng = np.random.default_rng(8)
    a = np.concatenate([datagen(lo, 250000, 50, 3), rng.integers(0, 256, size=40000, dtype=np.uint8), np.zeros(120000, np.uint8), text_like(150000, 6)])
    cp = _cp(lo, level, len(a))
    got = emu_compress_frame_jobs_lazy(le, lo, a, (C.c_uint * 7)(*cp), row, js, ov, bool(ck))
    lo.zo_set_row_matcher(1 if row else 0)
    try:
        want = oracle_frame_mt(lo, a, 0, js, ov, ck, cp=(C.c_uint * 7)(*cp))
    finally:
        lo.zo_set_row_matcher(0)
    assert got == want, (level, row, js, ov, ck)


def test_lazy_frames_with_the_two_pass_prediction(libs, monkeypatch):
    """$ZHIP_LZ_PREDICT=1: k_lz_predict marks what the parse will leave un-inserted, k_lz_search runs again without those positions, the exact
    parse only distrusts records where prediction and truth differ — same frames"""
    lo, le = libs
    monkeypatch.setenv("ZHIP_LZ_PREDICT", "1")
    cases = _cases(lo)
    bufs = [a for _, a in cases]
    for level, row in ((5, 1),):
        cps = [_cp(lo, level, len(a)) for a in bufs]
        got = emu_compress_frames_lazy(le, lo, bufs, cps, row)
        for (name, a), cp, g in zip(cases, cps, got):
            assert g == oracle_frame_params(lo, a, (C.c_uint * 7)(*cp), row), (name, level, row)


@pytest.mark.parametrize("ring,predict", [("0", "0"), ("1", "0")])            # ("0", "1"): tests/test_gpu_frames_lazy.py::test_live_rows_switch_gives_the_same_bytes
def test_lazy_frames_without_the_live_rows_and_without_the_prediction(libs, monkeypatch, ring, predict):
    """the other forms of the exact parse (the suite's default is live rows + probed prediction): live searches that walk the links ($ZHIP_LZ_RING=0 —
    also what a context falls back to when the rows' arena cannot be allocated), with and without the two-pass prediction; and one parse from the
    live rows alone.  Long-match data (the probe decides to predict), text (it does not) and a frame with both — the same frames as the oracle's."""
    lo, le = libs
    monkeypatch.setenv("ZHIP_LZ_RING", ring)
    monkeypatch.setenv("ZHIP_LZ_PREDICT", predict)
    bufs = [datagen(lo, 160000, 50, 11), text_like(135000, 3), np.concatenate([text_like(70000, 5), datagen(lo, 100000, 80, 6)])]
    cps = [_cp(lo, 5, len(a)) for a in bufs]
    got = emu_compress_frames_lazy(le, lo, bufs, cps, 1)
    for a, cp, g in zip(bufs, cps, got):
        assert g == oracle_frame_params(lo, a, (C.c_uint * 7)(*cp), 1), (ring, predict, len(a))
