"""-m gpu: stage-1 parity on a real MI355X through the C ABI (zhip_parse_device / zhip_get_sequences) against the
oracle's ZSTD_generateSequences-format output (oracle/zoracle.c, pinned to the reference)."""
import ctypes as C
import numpy as np
import pytest
from _libs import load_oracle, corpus_cases, datagen, _buf, ERR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import zstd_amd
    assert torch.cuda.is_available(), "needs a GPU"
    return load_oracle(), zstd_amd.Context(0, max_units=64), torch


def oracle_public(lo, a, level):
    n = len(a)
    cp = (C.c_uint * 7)()
    assert lo.zo_get_cparams(level, n, cp) == 0
    cap = n // 3 + 8
    out = np.zeros((cap, 4), dtype=np.uint32)
    k = lo.zo_sequences_public(cp, _buf(a), n, _buf(out), cap)
    assert k != ERR
    return out[:k]


def gpu_seqs(ctx, torch, bufs, level, unit):
    flat = np.concatenate(bufs) if len(bufs) else np.zeros(0, np.uint8)
    d = torch.from_numpy(np.concatenate([flat, np.zeros(64, np.uint8)])).cuda()
    nu = ctx.parse_device(d.data_ptr(), len(flat), level, unit)
    assert nu == max(1, len(bufs))
    return [ctx.get_sequences(i) for i in range(len(bufs))]


@pytest.mark.parametrize("level", [1, 3])
def test_parse_128k_units_match_oracle(env, level):
    lo, ctx, torch = env
    cases = list(corpus_cases(lo, sizes=(131072,), seeds=(0, 1)))
    res = gpu_seqs(ctx, torch, [c[1] for c in cases], level, 131072)
    for (name, a), s in zip(cases, res):
        o = oracle_public(lo, a, level)
        assert s.shape == o.shape and np.array_equal(s, o), name


@pytest.mark.parametrize("level", [1, -1, 2, 3, 4])
def test_parse_ragged_units_match_oracle(env, level):
    lo, ctx, torch = env
    import zstd_amd
    for n in (1, 7, 13, 14, 100, 257, 1000, 5000, 16384, 40000, 100001):
        try:
            cp = zstd_amd.get_cparams(level, n)
        except zstd_amd.ZhipError:
            continue
        if cp[6] not in (1, 2):
            continue
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(4,)):
            s = gpu_seqs(ctx, torch, [a], level, 131072)[0]
            o = oracle_public(lo, a, level)
            assert s.shape == o.shape and np.array_equal(s, o), (name, level)


def test_parse_many_units_tail(env):
    lo, ctx, torch = env
    n = 131072 * 5 + 777
    a = datagen(lo, n, 50, 9)
    d = torch.from_numpy(np.concatenate([a, np.zeros(64, np.uint8)])).cuda()
    nu = ctx.parse_device(d.data_ptr(), n, 1, 131072)
    assert nu == 6
    for i in range(6):
        chunk = a[i * 131072: min(n, (i + 1) * 131072)]
        assert np.array_equal(ctx.get_sequences(i), oracle_public(lo, chunk, 1)), i
