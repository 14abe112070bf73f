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
            if got == ERR:      # a strategy whose dictionary path is not restated
                assert False, (kind, level, len(r))
                continue
            assert mine[:got].tobytes() == want, (kind, level, dsize, len(r), got, cs)
            checked += 1
        assert checked >= len(recs) // 2
        lo.zo_cdict_free(cd)


def train_zdict(lr, samples, cap=112640):
    lr.zref_train_dict.restype = C.c_size_t
    lr.zref_train_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint]
    flat = np.concatenate(samples)
    sizes = (C.c_size_t * len(samples))(*[len(s) for s in samples])
    buf = np.zeros(cap, dtype=np.uint8)
    r = lr.zref_train_dict(_buf(buf), cap, _buf(flat), sizes, len(samples))
    assert r != ERR, "ZDICT_trainFromBuffer failed"
    return buf[:r].copy()


@pytest.mark.parametrize("level", [1, 3, 4])
@pytest.mark.parametrize("kind", ["json", "text"])
def test_zdict_trained_dictionary_records_match_the_reference(level, kind):
    """ZDICT-format dictionary (entropy tables + repcodes + content): repeat-mode literals / FSE tables, byte for byte"""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from zstd_amd import workloads as W
    lo, lr = load_oracle(), load_ref()
    bind(lo, lr)
    rng = np.random.default_rng(level)
    if kind == "json":
        flat, offs = W.github_like_records(3000, seed=level)
        recs = [flat[int(offs[i]):int(offs[i + 1])].copy() for i in range(len(offs) - 1)]
    else:
        corpus = text_like(600000, 5)
        recs = [corpus[s:s + int(n)].copy() for s, n in zip(rng.integers(0, 590000, size=3000), rng.integers(50, 3000, size=3000))]
    for cap in (112640, 20000):
        zd = train_zdict(lr, recs[:2000], cap)
        assert zd[:4].tobytes() == (0xEC30A437).to_bytes(4, "little")
        cd = lo.zo_cdict_create(_buf(zd), len(zd), level)
        assert cd
        test = recs[2000:2300] + [np.zeros(0, np.uint8), recs[0][:5], recs[1][:7], recs[2][:8], recs[3][:20], recs[4][:64], np.concatenate(recs[5:11])[:8000],
                                  rng.integers(0, 256, size=1500, dtype=np.uint8), np.full(900, 65, np.uint8)]
        flat2 = np.concatenate(test + [np.zeros(8, np.uint8)])
        sizes = (C.c_size_t * len(test))(*[len(r) for r in test])
        capd = sum(len(r) + 64 for r in test) + 4096
        dst = np.zeros(capd, dtype=np.uint8)
        osz = (C.c_size_t * len(test))()
        tot = lr.zref_compress_records_cdict(level, _buf(zd), len(zd), _buf(flat2), sizes, len(test), _buf(dst), capd, osz)
        assert tot != ERR
        pos = 0
        for i, (r, cs) in enumerate(zip(test, osz)):
            want = dst[pos:pos + cs].tobytes()
            pos += cs
            mine = np.zeros(len(r) + 600, dtype=np.uint8)
            got = lo.zo_compress_unit_cdict(_buf(mine), len(mine), _buf(r), len(r), cd)
            assert got != ERR
            assert mine[:got].tobytes() == want, (kind, level, cap, i, len(r), got, cs, mine[:12].tobytes().hex(), want[:12].hex())
        lo.zo_cdict_free(cd)


def test_fuzzed_dictionaries_and_records_match_the_reference():
    """structured-random dictionaries / records (tests/test_fuzz_emu.gen), raw content, levels -3..4"""
    import os
    from test_fuzz_emu import gen
    lo, lr = load_oracle(), load_ref()
    bind(lo, lr)
    rounds = int(os.environ.get("ZHIP_DICT_FUZZ_ROUNDS", "40"))
    for rd in range(rounds):
        rng = np.random.default_rng(4000 + rd)
        level = (-3, 1, 2, 3, 4)[rd % 5]
        d = gen(rng, int(rng.integers(8, 70000)))
        cd = lo.zo_cdict_create(_buf(d), len(d), level)
        if not cd:
            continue
        recs = []
        for _ in range(10):
            n = int(rng.integers(0, 5000))
            r = gen(rng, n)
            for _ in range(int(rng.integers(0, 6))):
                ln = min(int(rng.integers(4, 200)), n, len(d))
                if ln == 0:
                    continue
                s = len(d) - ln if rng.random() < 0.2 else int(rng.integers(0, len(d) - ln + 1))
                o = int(rng.integers(0, n - ln + 1))
                r[o:o + ln] = d[s:s + ln]
            recs.append(r)
        flat = np.concatenate(recs + [np.zeros(8, np.uint8)])
        sizes = (C.c_size_t * len(recs))(*[len(r) for r in recs])
        cap = sum(len(r) + 64 for r in recs) + 4096
        dst = np.zeros(cap, dtype=np.uint8)
        osz = (C.c_size_t * len(recs))()
        tot = lr.zref_compress_records_cdict(level, _buf(d), len(d), _buf(flat), sizes, len(recs), _buf(dst), cap, osz)
        assert tot != ERR
        pos = 0
        for r, cs in zip(recs, osz):
            want = dst[pos:pos + cs].tobytes(); pos += cs
            mine = np.zeros(len(r) + 600, dtype=np.uint8)
            got = lo.zo_compress_unit_cdict(_buf(mine), len(mine), _buf(r), len(r), cd)
            assert got != ERR and mine[:got].tobytes() == want, (rd, level, len(r), len(d))
        lo.zo_cdict_free(cd)


def test_copy_mode_records_above_the_attach_cutoff_match_the_reference():
    """records above the attach cut-off (8 KB fast / 16 KB dfast): the reference copies the CDict's tables and runs the
    _extDict block compressors (zstd_compress.c:2395, zstd_fast.c:709, zstd_double_fast.c:551)"""
    import os
    from test_fuzz_emu import gen
    lo, lr = load_oracle(), load_ref()
    bind(lo, lr)
    zd = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "github_like_110k.zdict"), dtype=np.uint8)
    for rd in range(int(os.environ.get("ZHIP_DICT_FUZZ_ROUNDS", "15"))):
        rng = np.random.default_rng(555 + rd)
        level = (3, 1, 4, 2, -3)[rd % 5]
        d = zd if rd % 4 == 0 else text_like(int(rng.integers(20000, 120000)), rd) if rd % 4 == 1 else gen(rng, int(rng.integers(9, 100000)))
        cd = lo.zo_cdict_create(_buf(d), len(d), level)
        if not cd:
            continue
        recs = []
        for _ in range(4):
            n = int(rng.integers(8193, 131073))
            r = (text_like(n, rd + 7) if rd % 2 == 0 else gen(rng, n)).copy()
            for _ in range(int(rng.integers(0, 30))):
                ln = min(int(rng.integers(8, 3000)), len(d), n)
                s0 = int(rng.integers(0, len(d) - ln + 1)); o = int(rng.integers(0, n - ln + 1))
                r[o:o + ln] = d[s0:s0 + ln]
            recs.append(r)
        flat = np.concatenate(recs + [np.zeros(8, np.uint8)])
        sizes = (C.c_size_t * len(recs))(*[len(r) for r in recs])
        cap = sum(len(r) + 700 for r in recs) + 4096
        dst = np.zeros(cap, dtype=np.uint8)
        osz = (C.c_size_t * len(recs))()
        tot = lr.zref_compress_records_cdict(level, _buf(d), len(d), _buf(flat), sizes, len(recs), _buf(dst), cap, osz)
        assert tot != ERR
        pos = 0
        for r, cs in zip(recs, osz):
            want = dst[pos:pos + cs].tobytes(); pos += cs
            mine = np.zeros(len(r) + 700, dtype=np.uint8)
            got = lo.zo_compress_unit_cdict(_buf(mine), len(mine), _buf(r), len(r), cd)
            assert got != ERR and mine[:got].tobytes() == want, (rd, level, len(r), len(d))
        lo.zo_cdict_free(cd)
