"""Dictionary path (CDict, attach mode): the oracle's restatement vs the REAL reference (oracle/_ref), byte for byte.
Needs /root/reference-built oracle/_ref (skipped where it is absent); tests/golden/dict_v1.json pins the same cases
for the GPU box."""
import ctypes as C
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, datagen, text_like, _buf, ERR

pytestmark = pytest.mark.skipif(not have_ref(), reason="reference build (oracle/_ref) not present")


def bind(lo, lr):
    lo.zo_cdict_create.restype = C.c_void_p
    lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lo.zo_cdict_free.argtypes = [C.c_void_p]
    lo.zo_compress_unit_cdict.restype = C.c_size_t
    lo.zo_compress_unit_cdict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lr.zref_compress_records_cdict.restype = C.c_size_t
    lr.zref_compress_records_cdict.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_size_t, C.c_void_p]
    lr.zref_decompress_dict.restype = C.c_size_t
    lr.zref_decompress_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]


def make_records(kind, seed, nrec=60):
    """records that share material with the dictionary: slices of one corpus, mutated"""
    rng = np.random.default_rng(seed)
    corpus = text_like(300000, seed) if kind == "text" else datagen(load_oracle(), 300000, 60, seed)
    dict_ = corpus[:110000 if kind == "text" else 40000].copy()
    recs = []
    for i in range(nrec):
        n = int(rng.choice([0, 1, 6, 7, 8, 9, 20, 100, 500, 1000, 1024, 2000, 4000, 8000, 8192, 12000, 16384]))
        s = int(rng.integers(0, len(corpus) - n))
        r = corpus[s:s + n].copy()
        if n > 50:
            k = rng.integers(0, n, size=n // 40)
            r[k] = rng.integers(0, 256, size=len(k), dtype=np.uint8)
        recs.append(r)
    return dict_, recs


@pytest.mark.parametrize("level", [1, 2, 3, 4, -1])
@pytest.mark.parametrize("kind", ["text", "datagen"])
def test_cdict_records_match_the_reference(level, kind):
    lo, lr = load_oracle(), load_ref()
    bind(lo, lr)
    dict_, recs = make_records(kind, 10 + level)
    for dsize in (len(dict_), 5000, 9, 7):
        d = dict_[:dsize].copy()
        cd = lo.zo_cdict_create(_buf(d), len(d), level)
        if not cd:      # the CDict's own row is a lazy strategy (e.g. level 4 with a dictionary below 16 KB): dictionary variants of
            assert level >= 4      # the hash-chain matchers are not restated
            continue
        flat = np.concatenate(recs + [np.zeros(8, np.uint8)])
        sizes = (C.c_size_t * len(recs))(*[len(r) for r in recs])
        cap = sum(len(r) + 64 for r in recs) + 4096
        dst = np.zeros(cap, dtype=np.uint8)
        osz = (C.c_size_t * len(recs))()
        tot = lr.zref_compress_records_cdict(level, _buf(d), len(d), _buf(flat), sizes, len(recs), _buf(dst), cap, osz)
        assert tot != ERR
        pos = 0
        checked = 0
        for r, cs in zip(recs, osz):
            want = dst[pos:pos + cs].tobytes()
            pos += cs
            mine = np.zeros(len(r) + 600, dtype=np.uint8)
            got = lo.zo_compress_unit_cdict(_buf(mine), len(mine), _buf(r), len(r), cd)
            if got == ERR:      # above the attach cutoff of this strategy: the reference copies the dictionary (not restated)
                assert len(r) > 8192
                continue
            assert mine[:got].tobytes() == want, (kind, level, dsize, len(r), got, cs)
            checked += 1
        assert checked >= len(recs) // 2
        lo.zo_cdict_free(cd)
