"""Row-hash match finder on the GPU: levels 5-10 with the reference's DEFAULT matcher selection, bytes against
the oracle's row matcher (pinned to the reference, fresh CCtx per unit), the committed golden vectors made from the real reference
(tests/golden/units_v3_rowhash.json) and, where oracle/_ref travels, the reference itself."""
import ctypes as C
import hashlib
import json
import os
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, corpus_cases, datagen, text_like, lorem, _buf, ERR, ROOT

pytestmark = pytest.mark.gpu
UNIT = 131072


@pytest.fixture(scope="module")
def env():
    import zstd_amd
    lo = load_oracle()
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    ctx = zstd_amd.Context(0, max_units=64)
    ctx.set_row_matcher(0)                       # auto = the reference's default (conftest makes hash chain the initial mode of the suite)
    return zstd_amd, ctx, lo


def test_rowhash_golden_vectors(env):
    zstd_amd, ctx, lo = env
    gold = {(g["case"], g["level"]): g for g in json.load(open(os.path.join(ROOT, "tests", "golden", "units_v3_rowhash.json")))["units"]}
    seen = 0
    for n in (131072, 100001, 40000, 20000):
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(0,)):
            for level in (5, 6, 7, 8, 9, 10):
                g = gold.get((name, level))
                if g is None:
                    continue
                got = ctx.compress(a, level=level)
                assert len(got) == g["csize"] and hashlib.sha256(got).hexdigest() == g["dst_sha256"], (name, level)
                seen += 1
    assert seen == len(gold)


def test_rowhash_batch_matches_oracle_and_reference(env):
    zstd_amd, ctx, lo = env
    a = np.concatenate([datagen(lo, 12 * UNIT + 30000, 50, 6), text_like(6 * UNIT, 4),
                        np.tile(np.random.default_rng(1).integers(0, 256, 900, dtype=np.uint8), 300)[: 2 * UNIT],     # long matches: the 384-position skip rule
                        np.repeat(np.random.default_rng(2).integers(0, 256, 2 * UNIT // 24 + 1, dtype=np.uint8), 24)[: 2 * UNIT]])   # runs of 24: every sequence a repcode, greedy never searches (round 6: was 4.4 s per unit)
    lr = load_ref() if have_ref() else None
    if lr is not None:
        lr.zref_compress_frame.restype = C.c_size_t
        lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lo.zo_set_row_matcher(1)
    try:
        for level in (5, 7, 8, 10):
            got = ctx.compress(a, level=level)
            cap = lo.zo_compress_bound(UNIT) * 26
            want = np.empty(cap, dtype=np.uint8)
            r = lo.zo_compress_chunks(level, UNIT, _buf(a), len(a), _buf(want), cap, None, 0)
            assert r != ERR and got == want[:r].tobytes(), level
            if lr is not None:
                ref = b""
                for off in range(0, len(a), UNIT):
                    u = a[off: off + UNIT]
                    d = np.zeros(len(u) + 1024, dtype=np.uint8)
                    k = lr.zref_compress_frame(level, _buf(u), len(u), _buf(d), len(d))
                    assert k != ERR
                    ref += d[:k].tobytes()
                assert got == ref, ("reference, fresh CCtx per unit", level)
            # and the hash-chain mode is still there
        ctx.set_row_matcher(2)
        lo.zo_set_row_matcher(0)
        got = ctx.compress(a, level=5)
        r = lo.zo_compress_chunks(5, UNIT, _buf(a), len(a), _buf(want), cap, None, 0)
        assert got == want[:r].tobytes()
        ctx.set_row_matcher(0)
    finally:
        lo.zo_set_row_matcher(0)


def test_lorem_ipsum_units_equal_the_reference(env):
    """the input of `zstd -b#` without a file — LOREM_genBuffer(.., seed 0), programs/benchzstd.c:1014, made by the reference's own generator — in 128 KB units:
    the device's frames are the reference's (fresh CCtx per unit) at the default level of every match-finder family"""
    zstd_amd, ctx, lo = env
    if not have_ref():
        pytest.skip("oracle/_ref (the reference build) did not travel")
    lr = load_ref()
    a = lorem(lr, 9 * UNIT + 4321, 0)
    assert bytes(a[:27]) == b"Lorem ipsum dolor sit amet,"
    for level in (1, 3, 5, 7, 8):
        got = ctx.compress(a, level=level)
        ref = b""
        for off in range(0, len(a), UNIT):
            u = np.ascontiguousarray(a[off: off + UNIT])
            d = np.zeros(len(u) + 1024, dtype=np.uint8)
            k = lr.zref_compress_frame(level, _buf(u), len(u), _buf(d), len(d))
            assert k != ERR
            ref += d[:k].tobytes()
        assert got == ref, level


def test_rowhash_frames_decode_on_the_device(env):
    zstd_amd, ctx, lo = env
    a = datagen(lo, 9 * UNIT + 5, 35, 2)
    for level in (5, 8):
        got = ctx.compress(a, level=level)
        assert zstd_amd.DContext(0).decompress(got) == a.tobytes()
