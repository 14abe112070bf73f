"""Pins oracle/zoracle.c (our C restatement) against the REAL reference built from /root/reference
(oracle/_ref).  CPU only.  Skipped when oracle/_ref is absent."""
import ctypes as C
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, corpus_cases, datagen, text_like, lorem, oracle_frame, oracle_frame_mt, ref_frame_mt, _buf, ERR
from _libs import ROOT, oracle_frame_params
import os

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def libs():
    return load_oracle(), load_ref()


def ref_unit(lr, a, level):
    cap = lr.zref_compress_bound(len(a)) + 64
    dst = np.zeros(cap, dtype=np.uint8)
    # hash-chain matcher for greedy/lazy/lazy2 (no effect on the fast/dfast levels)
    r = lr.zref_compress_chunks_norow(level, 1 << 17, _buf(a), len(a), _buf(dst), cap, None, 0)
    assert r != ERR
    return dst[:r].tobytes()


def ora_unit(lo, a, level):
    cap = lo.zo_compress_bound(len(a)) + 64
    dst = np.zeros(cap, dtype=np.uint8)
    r = lo.zo_compress_unit(_buf(dst), cap, _buf(a), len(a), level)
    assert r != ERR
    return dst[:r].tobytes()


def test_datagen_matches_reference(libs):
    lo, lr = libs
    for P in (0, 20, 50, 90, 100):
        for seed in (0, 3):
            for n in (0, 1, 1000, 200000):
                a = datagen(lo, n, P, seed)
                b = np.zeros(max(n, 1), dtype=np.uint8)
                lr.zref_datagen(_buf(b), n, P / 100.0, 0.0, seed)
                assert a.tobytes() == b[:n].tobytes(), (P, seed, n)


def test_cparams_match_reference(libs):
    lo, lr = libs
    sizes = [1, 5, 63, 64, 65, 100, 255, 256, 257, 511, 512, 513, 1000, 1024, 4095, 4096, 16383, 16384, 16385,
             65536, 131071, 131072, 131073, 262144, 262145, 1 << 20, 1 << 30, (1 << 30) + 1, 1 << 32]
    for level in (-5, -1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12):
        for n in sizes:
            o = (C.c_uint * 7)()
            r = (C.c_int * 7)()
            rc = lo.zo_get_cparams(level, n, o)
            lr.zref_get_cparams(level, n, 0, r)
            if r[6] > 5:            # binary-tree strategies are out of scope for the oracle
                assert rc == -1
                continue
            assert rc == 0 and list(o) == list(r), (level, n, list(o), list(r))


@pytest.mark.parametrize("level", [1, 3, 5, 6, 7, 10])
def test_unit_bytes_match_reference_128k(libs, level):
    lo, lr = libs
    for name, a in corpus_cases(lo, sizes=(131072,), seeds=(0, 1)):
        assert ora_unit(lo, a, level) == ref_unit(lr, a, level), name


@pytest.mark.parametrize("level", [1, 2, 3, 4, -1, -3, 5, 6, 8])
def test_unit_bytes_match_reference_small_and_ragged(libs, level):
    lo, lr = libs
    sizes = [0, 1, 2, 6, 7, 8, 9, 12, 15, 16, 17, 31, 32, 63, 64, 65, 100, 255, 256, 257, 300, 1000, 1023, 1024, 1025,
             4095, 4096, 5000, 16383, 16384, 16385, 40000, 65535, 65536, 65537, 100000, 131071]
    for n in sizes:
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(0,)):
            o = (C.c_uint * 7)()
            if lo.zo_get_cparams(level, n, o) != 0:
                continue
            assert ora_unit(lo, a, level) == ref_unit(lr, a, level), (name, level)


@pytest.mark.parametrize("level", [1, 3, 5, 6, 7])
def test_sequences_match_reference(libs, level):
    lo, lr = libs
    for name, a in corpus_cases(lo, sizes=(131072, 30000), seeds=(2,)):
        n = len(a)
        cap = n // 3 + 8
        so = np.zeros(cap * 4, dtype=np.uint32)
        sr = np.zeros(cap * 4, dtype=np.uint32)
        cp = (C.c_uint * 7)()
        assert lo.zo_get_cparams(level, n, cp) == 0
        no = lo.zo_sequences_public(cp, _buf(a), n, _buf(so), cap)
        nr = lr.zref_sequences(level, _buf(a), n, _buf(sr), cap)
        assert no != ERR and nr != ERR
        assert no == nr, name
        sr[4 * nr - 1] = 0      # the reference leaves the delimiter's .rep uninitialised (zstd_compress.c:3445-3447)
        assert np.array_equal(so[: 4 * no], sr[: 4 * nr]), name


def test_chunks_stream_matches_reference_and_roundtrips(libs):
    lo, lr = libs
    n = 1_000_003
    a = datagen(lo, n, 50, 11)
    cap = lr.zref_compress_bound(131072) * (n // 131072 + 1)
    do = np.zeros(cap, dtype=np.uint8)
    dr = np.zeros(cap, dtype=np.uint8)
    ro = lo.zo_compress_chunks(1, 131072, _buf(a), n, _buf(do), cap, None, 0)
    rr = lr.zref_compress_chunks(1, 131072, _buf(a), n, _buf(dr), cap, None, 0)
    assert ro == rr and do[:ro].tobytes() == dr[:rr].tobytes()
    out = np.zeros(n, dtype=np.uint8)
    assert lr.zref_decompress(_buf(out), n, _buf(do), ro) == n
    assert out.tobytes() == a.tobytes()


def test_stage_huffman_lengths_match_reference(libs):
    lo, lr = libs
    rng = np.random.default_rng(5)
    for trial in range(300):
        k = int(rng.integers(2, 257))
        kind = trial % 4
        if kind == 0:
            cnt = rng.integers(1, 50, size=k)
        elif kind == 1:
            cnt = (rng.pareto(0.7, size=k) * 20 + 1).astype(np.int64)
        elif kind == 2:
            cnt = rng.integers(150, 700, size=k)          # straddles the 166 bucket cutoff + log2 buckets
        else:
            cnt = (2.0 ** rng.uniform(0, 16, size=k)).astype(np.int64)
        cnt = np.minimum(cnt, 100000)
        if rng.random() < 0.5:
            cnt[rng.integers(0, k, size=k // 3)] = 0
        cnt[k - 1] = max(cnt[k - 1], 1)
        if (cnt > 0).sum() < 2:
            cnt[0] = 3
        count = np.zeros(256, dtype=np.uint32)
        count[:k] = cnt
        for maxbits in (11, 9, 8):
            if (cnt > 0).sum() > (1 << maxbits):
                continue
            bo = np.zeros(256, dtype=np.uint8)
            br = np.zeros(256, dtype=np.uint8)
            to = lo.zo_huf_build(_buf(count), k - 1, maxbits, _buf(bo))
            tr = lr.zref_huf_build(_buf(count), k - 1, maxbits, _buf(br))
            assert tr != ERR
            assert to == tr and np.array_equal(bo[:k], br[:k]), (trial, maxbits)


def test_stage_fse_normalize_matches_reference(libs):
    lo, lr = libs
    rng = np.random.default_rng(6)
    for trial in range(400):
        maxsym = int(rng.integers(1, 53))
        cnt = (rng.pareto(0.8, size=maxsym + 1) * 3).astype(np.int64)
        cnt[maxsym] = max(cnt[maxsym], 1)
        cnt[0] = max(cnt[0], 1)
        total = int(cnt.sum())
        if total < 2:
            continue
        count = np.zeros(64, dtype=np.uint32)
        count[: maxsym + 1] = cnt
        tl = lr.zref_fse_optimal_tablelog(9, total, maxsym)
        for low in (0, 1):
            no = np.zeros(64, dtype=np.int16)
            nr = np.zeros(64, dtype=np.int16)
            ro = lo.zo_fse_normalize(_buf(no), tl, _buf(count), total, maxsym, low)
            rr = lr.zref_fse_normalize(_buf(nr), tl, _buf(count), total, maxsym, low)
            if rr == ERR:
                assert ro == -1
                continue
            assert ro == rr and np.array_equal(no[: maxsym + 1], nr[: maxsym + 1]), (trial, low)


@pytest.mark.parametrize("level", [5, 6, 7, 8, 9, 10])
def test_rowhash_unit_bytes_match_reference_fresh_cctx(libs, level):
    """the reference's DEFAULT matcher for greedy / lazy / lazy2 (row hash, windowLog > 14): ZSTD_compress2 on a FRESH CCtx per unit
    (its hash salt is then the constant a new context starts with, zstd_compress.c:1964-1975, :2027-2033)"""
    lo, lr = libs
    lr.zref_compress_frame.restype = C.c_size_t
    lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    lo.zo_set_row_matcher(1)
    try:
        for n in (131072, 70001, 20000, 16385, 16384):
            for name, a in corpus_cases(lo, sizes=(n,), seeds=(0, 3)):
                o = (C.c_uint * 7)()
                if lo.zo_get_cparams(level, n, o) != 0:
                    continue
                want = np.zeros(n + 1024, dtype=np.uint8)
                k = lr.zref_compress_frame(level, _buf(a), n, _buf(want), len(want))
                assert k != ERR and ora_unit(lo, a, level) == want[:k].tobytes(), (name, level)
    finally:
        lo.zo_set_row_matcher(0)


def test_lorem_ipsum_units_vs_reference(libs):
    """`zstd -b#` without a file benches LOREM_genBuffer(.., seed 0) (programs/benchzstd.c:1014): the oracle equals the reference on that text too, at the
    default level of every match-finder family (fast, dfast, greedy / lazy / lazy2 with the row matcher, fresh CCtx per unit)"""
    lo, lr = libs
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    a = lorem(lr, 3 * 131072 + 4321, 0)
    assert bytes(a[:27]) == b"Lorem ipsum dolor sit amet,"
    lo.zo_set_row_matcher(1)
    try:
        for level in (1, 3, 5, 7, 8):
            for off in range(0, len(a), 131072):
                u = np.ascontiguousarray(a[off: off + 131072])
                want = np.zeros(len(u) + 1024, dtype=np.uint8)
                k = lr.zref_compress_frame(level, _buf(u), len(u), _buf(want), len(want))
                assert k != ERR and ora_unit(lo, u, level) == want[:k].tobytes(), (level, off)
    finally:
        lo.zo_set_row_matcher(0)


def test_multiblock_frame_vs_reference(libs):
    """zo_compress_frame against ZSTD_compress2 of the whole input on a fresh CCtx (the frame the shim's single-frame mode must equal)"""
    lo, lr = libs
    lr.zref_compress_frame.restype = C.c_size_t
    lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(123)
    for trial in range(12):
        n = int(rng.integers(131073, 900000))
        kind = trial % 4
        a = (datagen(lo, n, int(rng.integers(10, 95)), trial) if kind == 0 else text_like(n, trial) if kind == 1 else
             np.concatenate([datagen(lo, n // 2, 60, trial), rng.integers(0, 256, size=n - n // 2, dtype=np.uint8)]) if kind == 2 else
             np.repeat(rng.integers(0, 256, size=n // 4096 + 1, dtype=np.uint8), 4096)[:n].copy())
        for level in (1, 2, 3, 4, -3):
            cp = (C.c_uint * 7)()
            assert lo.zo_get_cparams(level, n, cp) == 0
            if cp[6] not in (1, 2):
                continue
            want = np.zeros(n + (n >> 7) + 1024, dtype=np.uint8)
            k = lr.zref_compress_frame(level, _buf(a), n, _buf(want), len(want))
            assert k != ERR and oracle_frame(lo, a, level) == want[:k].tobytes(), (trial, kind, n, level)


def test_multiblock_frame_with_explicit_parameters_vs_reference(libs):
    """the frame loop with explicit parameters (a window smaller than the input, other table sizes / hash widths, dfast) against
    ZSTD_compress2 of the whole input on a CCtx with the same parameters set"""
    lo, lr = libs
    lr.zref_compress_chunks_level_params.restype = C.c_size_t
    lr.zref_compress_chunks_level_params.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    import zstd_amd                                            # the product's host-side parameter logic (no GPU needed)
    L = zstd_amd.lib()
    L.zhip_getCParams_explicit.restype = C.c_int
    L.zhip_getCParams_explicit.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p]
    lo.zo_compress_frame_params.restype = C.c_size_t
    lo.zo_compress_frame_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_frame_bound.restype = C.c_size_t
    lo.zo_frame_bound.argtypes = [C.c_size_t]
    a = datagen(lo, 900_000, 70, 33).copy()
    a[600_000:700_000] = a[100_000:200_000]
    b = text_like(500_000, 8)
    for src in (a, b):
        n = src.size
        for level, cp in ((1, [17, 0, 0, 0, 0, 0, 0]), (1, [18, 0, 15, 0, 4, 0, 0]), (3, [17, 15, 16, 0, 0, 0, 0]), (1, [0, 0, 13, 0, 6, 2, 0]), (2, [0, 0, 0, 0, 0, 0, 0])):
            eff = (C.c_uint * 7)()
            req = (C.c_uint * 7)(*cp)
            assert L.zhip_getCParams_explicit(level, n, req, eff) == 0
            if eff[6] not in (1, 2):
                continue
            cap = lo.zo_frame_bound(n)
            got = np.zeros(cap, dtype=np.uint8)
            r = lo.zo_compress_frame_params(_buf(got), cap, _buf(src), n, eff)
            want = np.zeros(cap + 1024, dtype=np.uint8)
            cpi = (C.c_int * 7)(*cp)
            k = lr.zref_compress_chunks_level_params(level, cpi, 0, n, _buf(src), n, _buf(want), len(want))
            assert r != ERR and k != ERR and got[:r].tobytes() == want[:k].tobytes(), (level, cp, list(eff))


def test_tiny_frames_vs_reference(libs):
    """frames of 0 .. 24 bytes: below 7 bytes the block is stored, at exactly 7 the fast / dfast parsers run with their search limit
    (end - 8) BEFORE the source — the reference compares pointers there; the oracle's index arithmetic wrapped and read far out of the
    buffer (found by tests/tools/emu_fuzz_frames.py, the kernel was right)"""
    lo, lr = libs
    lr.zref_compress_frame.restype = C.c_size_t
    lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lr.zref_compress_chunks_level_params.restype = C.c_size_t
    lr.zref_compress_chunks_level_params.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    import zstd_amd
    L = zstd_amd.lib()
    L.zhip_getCParams_explicit.restype = C.c_int
    L.zhip_getCParams_explicit.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(77)
    want = np.zeros(4096, dtype=np.uint8)
    for n in range(0, 25):
        for kind in range(3):
            a = (rng.integers(0, 256, size=n, dtype=np.uint8) if kind == 0 else np.full(n, 65, np.uint8) if kind == 1 else
                 np.tile(rng.integers(0, 256, size=3, dtype=np.uint8), 9)[:n].copy())
            for level in (1, 3, -5):
                k = lr.zref_compress_frame(level, _buf(a), n, _buf(want), len(want))
                assert k != ERR and oracle_frame(lo, a, level) == want[:k].tobytes(), (n, kind, level)
            for level, cp in ((1, [17, 13, 17, 1, 7, 16, 1]), (3, [18, 12, 12, 1, 5, 0, 2]), (1, [17, 0, 0, 0, 3, 0, 1])):
                eff = (C.c_uint * 7)()
                assert L.zhip_getCParams_explicit(level, max(n, 1), (C.c_uint * 7)(*cp), eff) == 0
                k = lr.zref_compress_chunks_level_params(level, (C.c_int * 7)(*cp), 0, max(n, 1), _buf(a), n, _buf(want), len(want))
                assert k != ERR and oracle_frame_params(lo, a, eff, False) == want[:k].tobytes(), (n, kind, level, cp, list(eff))


def test_frame_sizes_around_the_parsers_limits_vs_reference(libs):
    """every strategy up to lazy2 (levels 1 .. 10, the default matcher) on frames of 0 .. 40 bytes and on last blocks of 1 .. 33 bytes after
    one or two full ones: the sizes at which a parser's search limit (end - 8, end - 16 with the row matcher) falls before the block"""
    lo, lr = libs
    lr.zref_compress_frame.restype = C.c_size_t
    lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(5)
    want = np.zeros(1 << 20, dtype=np.uint8)
    seen = 0
    for n in list(range(0, 41, 1)) + [131072 + k for k in (1, 6, 7, 8, 15, 16, 17, 33)] + [262144 + 7, 262144 + 16]:
        kinds = range(3) if n < 64 else range(2, 3)
        for kind in kinds:
            a = (rng.integers(0, 256, size=n, dtype=np.uint8) if kind == 0 else np.full(n, 65, np.uint8) if kind == 1 else text_like(max(n, 1), n)[:n].copy())
            for level in ((1, 3, 5, 6, 8, 10) if n < 64 else (3, 5, 8, 10)):
                cp = (C.c_uint * 7)()
                if lo.zo_get_cparams(level, max(n, 1), cp) != 0 or cp[6] > 5:      # a bt* row of the small-size tables: out of scope
                    continue
                k = lr.zref_compress_frame(level, _buf(a), n, _buf(want), len(want))
                assert k != ERR and oracle_frame_params(lo, a, cp, cp[6] >= 3 and cp[0] > 14) == want[:k].tobytes(), (n, kind, level, list(cp))
                seen += 1
    assert seen >= 600


def test_job_pool_frame_vs_reference(libs):
    """zo_compress_frame_mt_params against ZSTD_compress2 with ZSTD_c_nbWorkers = 1 on random sizes, job sizes and overlaps (the frame
    the shim's nbWorkers mode must equal); at or below 512 KB the reference drops the workers"""
    lo, lr = libs
    rng = np.random.default_rng(321)
    for trial in range(16):
        n = int(rng.integers(400_000, 3_000_000))
        kind = trial % 4
        a = (datagen(lo, n, int(rng.integers(10, 95)), trial) if kind == 0 else text_like(n, trial) if kind == 1 else
             np.concatenate([datagen(lo, n // 2, 60, trial), rng.integers(0, 256, size=n - n // 2, dtype=np.uint8)]) if kind == 2 else
             np.repeat(rng.integers(0, 256, size=n // 4096 + 1, dtype=np.uint8), 4096)[:n].copy())
        for level in (1, 3, -3):
            js = int(rng.choice([0, 1, 524288, 600_001, 1 << 20, 1 << 21]))
            ov = int(rng.integers(0, 10))
            ck = bool(rng.integers(0, 2))
            assert oracle_frame_mt(lo, a, level, js, ov, ck) == ref_frame_mt(lr, a, level, js, ov, ck), (trial, kind, n, level, js, ov, ck)


def mt_explicit_cases(lo, trials, seed):
    """random inputs x explicit parameters x job sizes x overlaps for the job-pool frame (windows smaller than a job, table logs 12-17,
    minMatch 4-7, both strategies): yields (input, level, requested[7], effective[7], jobSize, overlapLog, checksum)"""
    import zstd_amd                                            # the product's host-side parameter logic (no GPU needed)
    L = zstd_amd.lib()
    L.zhip_getCParams_explicit.restype = C.c_int
    L.zhip_getCParams_explicit.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(seed)
    for trial in range(trials):
        n = int(rng.integers(530_000, 1_400_000))
        kind = trial % 4
        a = (datagen(lo, n, int(rng.integers(10, 95)), trial) if kind == 0 else text_like(n, trial) if kind == 1 else
             np.concatenate([datagen(lo, n // 2, 60, trial), rng.integers(0, 256, size=n - n // 2, dtype=np.uint8)]) if kind == 2 else
             np.repeat(rng.integers(0, 256, size=n // 512 + 1, dtype=np.uint8), 512)[:n].copy())
        level = int(rng.choice([1, 2, 3, -2]))
        req = [int(rng.choice([0, 17, 18, 19, 20])), int(rng.choice([0, 12, 15, 16])), int(rng.choice([0, 12, 13, 14, 15, 16, 17])), 0,
               int(rng.choice([0, 4, 5, 6, 7])), int(rng.choice([0, 0, 2, 8])), int(rng.choice([0, 1, 2]))]
        eff = (C.c_uint * 7)()
        if L.zhip_getCParams_explicit(level, n, (C.c_uint * 7)(*req), eff) != 0 or eff[6] not in (1, 2) or eff[0] < 17:
            continue
        yield a, level, req, eff, int(rng.choice([0, 524288, 600_001, 1 << 20])), int(rng.integers(0, 10)), bool(rng.integers(0, 2))


def test_job_pool_frame_with_explicit_parameters_vs_reference(libs):
    lo, lr = libs
    seen = 0
    for a, level, req, eff, js, ov, ck in mt_explicit_cases(lo, 40, 7):
        assert oracle_frame_mt(lo, a, level, js, ov, ck, cp=eff) == ref_frame_mt(lr, a, level, js, ov, ck, cp=req), (len(a), level, req, js, ov, ck)
        seen += 1
    assert seen >= 24


def test_lazy_multiblock_frames_vs_reference(libs):
    """zo_compress_frame_params at the greedy / lazy / lazy2 strategies against ZSTD_compress2 of the whole input: random sizes up to 2 MB,
    random explicit parameters (windows below the input size, chain tables far smaller than the window, search depths, minMatch 3-6),
    row matcher on and off"""
    lo, lr = libs
    from _libs import oracle_frame_params
    import zstd_amd
    L = zstd_amd.lib()
    L.zhip_getCParams_explicit.restype = C.c_int
    L.zhip_getCParams_explicit.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p, C.c_void_p]
    lr.zref_compress_chunks_level_params.restype = C.c_size_t
    lr.zref_compress_chunks_level_params.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(77)
    seen = 0
    for t in range(24):
        n = int(rng.integers(131073, 2_000_000))
        kind = t % 4
        a = (datagen(lo, n, int(rng.integers(10, 95)), t) if kind == 0 else text_like(n, t) if kind == 1 else
             np.concatenate([datagen(lo, n // 2, 60, t), rng.integers(0, 256, size=n - n // 2, dtype=np.uint8)]) if kind == 2 else
             np.repeat(rng.integers(0, 256, size=n // 4096 + 1, dtype=np.uint8), 4096)[:n].copy())
        level = int(rng.choice([5, 6, 7, 8, 10]))
        req = [int(rng.choice([0, 17, 18, 19, 20])), int(rng.choice([0, 0, 10, 14, 16])), int(rng.choice([0, 0, 12, 15, 17])), int(rng.choice([0, 0, 1, 3, 5, 6])),
               int(rng.choice([0, 0, 3, 4, 5, 6])), int(rng.choice([0, 0, 4, 32])), int(rng.choice([0, 3, 4, 5]))]
        eff = (C.c_uint * 7)()
        if L.zhip_getCParams_explicit(level, n, (C.c_uint * 7)(*req), eff) != 0 or eff[6] not in (3, 4, 5):
            continue
        no_row = int(rng.integers(0, 2))
        want = np.zeros(n + (n >> 7) + 1024, dtype=np.uint8)
        k = lr.zref_compress_chunks_level_params(level, (C.c_int * 7)(*req), no_row, n, _buf(a), n, _buf(want), len(want))
        assert k != ERR and oracle_frame_params(lo, a, eff, row=not no_row) == want[:k].tobytes(), (t, n, level, req, list(eff), no_row)
        seen += 1
    assert seen >= 12


def test_job_pool_frame_lazy_strategies_vs_reference(libs):
    """ZSTD_c_nbWorkers = 1 at the greedy / lazy / lazy2 levels: a later job's fresh context loads its prefix completely (every position
    up to 8 before its end, zstd_compress.c:4920-4964) into the hash chain or the rows, overlapLog 7 by default for lazy2 — the oracle
    against the reference, row matcher on and off (oracle-only so far: DESIGN.md §9 item 4)"""
    lo, lr = libs
    lr.zref_compress_frame_mt_norow.restype = C.c_size_t
    lr.zref_compress_frame_mt_norow.argtypes = [C.c_int, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(17)
    try:
        for t in range(6):
            n = int(rng.integers(600_000, 2_500_000))
            kind = t % 3
            a = (datagen(lo, n, int(rng.integers(10, 95)), t) if kind == 0 else text_like(n, t) if kind == 1 else
                 np.concatenate([datagen(lo, n // 2, 60, t), rng.integers(0, 256, size=n - n // 2, dtype=np.uint8)]))
            for level in (5, 7, 9):
                no_row = int(rng.integers(0, 2))
                js, ov, ck = int(rng.choice([0, 524288, 1 << 20])), int(rng.integers(0, 10)), bool(rng.integers(0, 2))
                lo.zo_set_row_matcher(0 if no_row else 1)
                want = np.zeros(n + (n >> 7) + 1024, dtype=np.uint8)
                k = lr.zref_compress_frame_mt_norow(level, None, js, ov, 1 if ck else 0, no_row, _buf(a), n, _buf(want), len(want))
                assert k != ERR and oracle_frame_mt(lo, a, level, js, ov, ck) == want[:k].tobytes(), (t, n, level, no_row, js, ov, ck)
    finally:
        lo.zo_set_row_matcher(0)


def test_lazy_cdict_records_vs_reference(libs):
    """CDicts at the greedy / lazy / lazy2 levels: attach mode up to 32 KB, copy mode above, row matcher and hash chain, random raw
    dictionaries and records — the oracle against ZSTD_createCDict_advanced2 + refCDict + compress2 on a fresh CCtx per record"""
    lo, lr = libs
    from _libs import oracle_records_cdict
    lr.zref_compress_records_cdict_fresh.restype = C.c_size_t
    lr.zref_compress_records_cdict_fresh.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(3)
    for t in range(3):
        dsz = int(rng.choice([3000, 20000, 60000, 112640]))
        corpus = text_like(400000, t) if t % 2 == 0 else np.concatenate([text_like(200000, t + 50), datagen(lo, 200000, 50, t)])
        d = corpus[:dsz].copy()
        recs = []
        for k in range(8):
            n = int(rng.choice([1, 8, 16, 40, 300, 1200, 5000, 20000, 32768, 32769, 70000]))
            o = int(rng.integers(dsz, len(corpus) - n))
            r = corpus[o:o + n].copy()
            if k % 3 == 0 and n > 20:
                r[n // 2: n // 2 + 5] = rng.integers(0, 256, 5)
            recs.append(r)
        src = np.concatenate(recs)
        sizes = (C.c_size_t * len(recs))(*[len(r) for r in recs])
        for level in (5, 7, 9):
            no_row = int(rng.integers(0, 2))
            cap = sum(len(r) + (len(r) >> 7) + 256 for r in recs)
            dst = np.zeros(cap, dtype=np.uint8)
            k = lr.zref_compress_records_cdict_fresh(level, no_row, _buf(d), len(d), _buf(src), sizes, len(recs), _buf(dst), cap, None)
            assert k != ERR and b"".join(oracle_records_cdict(lo, d, recs, level, row=not no_row)) == dst[:k].tobytes(), (t, dsz, level, no_row)


def test_cdict_on_sources_above_128k_vs_reference(libs):
    """ZSTD_createCDict + refCDict + ZSTD_compress2 on a source above 128 KB (strategies fast / dfast): while the source is below six times
    the dictionary (zstd_compress.c:5153-5190) the CDict's tables are copied and carried through the frame's blocks — every block runs the
    extDict parser until the window has slid past the dictionary, then the plain one; the first block starts from the dictionary's
    repcodes and entropy tables.  Larger sources make the reference reload the dictionary content with the context's own parameters:
    zo_compress_frame_cdict refuses those (not restated)"""
    lo, lr = libs
    lo.zo_cdict_create.restype = C.c_void_p; lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lo.zo_cdict_free.argtypes = [C.c_void_p]
    lo.zo_compress_frame_cdict.restype = C.c_size_t; lo.zo_compress_frame_cdict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_frame_bound.restype = C.c_size_t; lo.zo_frame_bound.argtypes = [C.c_size_t]
    lr.zref_compress_records_cdict.restype = C.c_size_t
    lr.zref_compress_records_cdict.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    zd = np.fromfile(os.path.join(ROOT, "tests", "golden", "github_like_110k.zdict"), dtype=np.uint8)
    rng = np.random.default_rng(9)
    seen = slid = 0
    for t in range(8):
        corpus = text_like(1300000, t) if t % 2 == 0 else np.concatenate([text_like(600000, t + 50), datagen(lo, 700000, 50, t)])
        d = zd if t % 3 == 2 else corpus[:int(rng.choice([60000, 112640, 200000]))].copy()
        n = int(rng.integers(131073, min(6 * len(d), 1200000)))
        o = int(rng.integers(0, len(corpus) - n))
        a = corpus[o:o + n].copy()
        for level in (1, 3, -3):
            cd = lo.zo_cdict_create(_buf(d), len(d), level)
            assert cd
            cap = lo.zo_frame_bound(n)
            out = np.zeros(cap, dtype=np.uint8)
            r = lo.zo_compress_frame_cdict(_buf(out), cap, _buf(a), n, cd)
            lo.zo_cdict_free(cd)
            assert r != ERR
            want = np.zeros(n + (n >> 7) + 1024, dtype=np.uint8)
            k = lr.zref_compress_records_cdict(level, _buf(d), len(d), _buf(a), (C.c_size_t * 1)(n), 1, _buf(want), len(want), None)
            assert k != ERR and out[:r].tobytes() == want[:k].tobytes(), (t, n, level, len(d))
            seen += 1
            slid += (level != 3 and n > (1 << 19))                     # levels 1 and -3: a 512 KB window, the dictionary slides out
    assert seen == 24 and slid >= 2
