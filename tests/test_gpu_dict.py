"""-m gpu: dictionary path (records compressed with an attached CDict) through the C ABI: every frame byte-identical to the
oracle's restatement of ZSTD_createCDict + ZSTD_CCtx_refCDict + ZSTD_compress2 (pinned to the real reference by
tests/test_oracle_dict.py), and decoded with the dictionary by the reference build when it is present."""
import ctypes as C
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, datagen, text_like, _buf, ERR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    import zstd_amd
    assert torch.cuda.is_available(), "needs a GPU"
    lo = load_oracle()
    lo.zo_cdict_create.restype = C.c_void_p
    lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lo.zo_cdict_free.argtypes = [C.c_void_p]
    lo.zo_compress_unit_cdict.restype = C.c_size_t
    lo.zo_compress_unit_cdict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    return lo, zstd_amd, torch


def records_of(corpus, rng, sizes):
    recs = []
    for n in sizes:
        s = int(rng.integers(0, len(corpus) - n))
        r = corpus[s:s + n].copy()
        if n > 50:
            k = rng.integers(0, n, size=max(1, n // 40))
            r[k] = rng.integers(0, 256, size=len(k), dtype=np.uint8)
        recs.append(r)
    return recs


def oracle_frames(lo, dict_, recs, level):
    cd = lo.zo_cdict_create(_buf(dict_), len(dict_), level)
    assert cd
    out = []
    for r in recs:
        buf = np.zeros(len(r) + 700, dtype=np.uint8)
        k = lo.zo_compress_unit_cdict(_buf(buf), len(buf), _buf(r), len(r), cd)
        assert k != ERR
        out.append(buf[:k].tobytes())
    lo.zo_cdict_free(cd)
    return out


@pytest.mark.parametrize("kind,level,dsize", [("text", 3, 110000), ("datagen", 3, 40000), ("text", 4, 60000), ("text", 3, 9), ("text", 3, 5),
                                              ("text", 1, 110000), ("datagen", 1, 30000), ("text", 2, 50000), ("text", -1, 20000), ("text", 1, 6)])
def test_records_with_dictionary_match_oracle_bytes(env, kind, level, dsize):
    lo, zstd_amd, torch = env
    rng = np.random.default_rng(abs(level) * 100 + dsize % 97)
    corpus = text_like(300000, 3) if kind == "text" else datagen(lo, 300000, 60, 3)
    dict_ = corpus[:dsize].copy()
    sizes = [0, 1, 6, 7, 8, 9, 10, 17, 64, 100, 300, 500, 1000, 1024, 1500, 2000, 4000, 8000, 8192] + [int(x) for x in rng.integers(200, 2000, size=200)]
    if zstd_amd.lib().zhip_getCParams(level, 200000, (C.c_uint * 7)()) == 0 and level >= 3:
        sizes += [12000, 16384]                               # strategy dfast attaches up to 16 KB, fast up to 8 KB
    recs = records_of(corpus, rng, sizes)
    recs.append(dict_[-300:].copy() if dsize > 300 else dict_.copy())
    ctx = zstd_amd.Context(0, max_units=len(recs), records_total_bytes=sum(len(r) for r in recs))
    cd = zstd_amd.CDict(dict_, level=level)
    got, fs = ctx.compress_records(cd, recs, return_sizes=True)
    want = oracle_frames(lo, dict_, recs, level)
    pos = 0
    for i, (r, w) in enumerate(zip(recs, want)):
        g = got[pos:pos + int(fs[i])]
        pos += int(fs[i])
        assert g == w, (kind, level, dsize, i, len(r), len(g), len(w))
    assert pos == len(got)
    if have_ref() and dsize >= 8:
        lr = load_ref()
        lr.zref_decompress_dict.restype = C.c_size_t
        lr.zref_decompress_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        pos = 0
        for i, r in enumerate(recs):
            f = np.frombuffer(got[pos:pos + int(fs[i])], dtype=np.uint8)
            pos += int(fs[i])
            out = np.zeros(max(1, len(r)), dtype=np.uint8)
            k = lr.zref_decompress_dict(_buf(out), len(r), _buf(f), len(f), _buf(dict_), len(dict_))
            assert k == len(r) and out[:len(r)].tobytes() == r.tobytes(), i
    cd.close(); ctx.close()


@pytest.mark.parametrize("level", [3, 4, 1])
def test_zdict_trained_dictionary_records_match_oracle_bytes(env, level):
    """ZDICT-format dictionary (tests/golden/github_like_110k.zdict, trained with the real reference): dictID in the frame
    header, repcodes from the dictionary, treeless literals and set_repeat FSE tables — byte-identical to the oracle"""
    import os, sys
    lo, zstd_amd, torch = env
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from zstd_amd import workloads as W
    zd = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "github_like_110k.zdict"), dtype=np.uint8)
    flat, offs = W.github_like_records(400, seed=7)
    recs = [flat[int(offs[i]):int(offs[i + 1])].copy() for i in range(400)]
    rng = np.random.default_rng(level)
    recs += [np.zeros(0, np.uint8), recs[0][:5], recs[1][:6], recs[2][:7], recs[3][:8], recs[4][:9], recs[5][:30], recs[6][:64], recs[7][:100],
             np.concatenate(recs[8:14])[:8000], np.concatenate(recs[20:40])[:16384 if level >= 3 else 8192], rng.integers(0, 256, size=1500, dtype=np.uint8),
             np.full(900, 65, np.uint8), np.full(7, 66, np.uint8), text_like(3000, 4)]
    ctx = zstd_amd.Context(0, max_units=len(recs), records_total_bytes=sum(len(r) for r in recs))
    cd = zstd_amd.CDict(zd, level=level)
    got, fs = ctx.compress_records(cd, recs, return_sizes=True)
    want = oracle_frames(lo, zd, recs, level)
    pos = 0
    kinds = set()
    for i, (r, w) in enumerate(zip(recs, want)):
        g = got[pos:pos + int(fs[i])]
        pos += int(fs[i])
        assert g == w, (level, i, len(r), len(g), len(w), g[:16].hex(), w[:16].hex())
    assert pos == len(got)
    if have_ref():
        lr = load_ref()
        lr.zref_decompress_dict.restype = C.c_size_t
        lr.zref_decompress_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        pos = 0
        for i, r in enumerate(recs):
            f = np.frombuffer(got[pos:pos + int(fs[i])], dtype=np.uint8)
            pos += int(fs[i])
            out = np.zeros(max(1, len(r)), dtype=np.uint8)
            k = lr.zref_decompress_dict(_buf(out), len(r), _buf(f), len(f), _buf(zd), len(zd))
            assert k == len(r) and out[:len(r)].tobytes() == r.tobytes(), i
    cd.close(); ctx.close()


def test_unsupported_dictionaries_are_errors(env):
    lo, zstd_amd, torch = env
    zd = np.zeros(200, dtype=np.uint8)
    zd[:4] = np.frombuffer((0xEC30A437).to_bytes(4, "little"), dtype=np.uint8)
    with pytest.raises(zstd_amd.ZhipError):
        zstd_amd.CDict(zd, level=3)                          # malformed ZDICT-format dictionary
    with pytest.raises(zstd_amd.ZhipError):
        zstd_amd.CDict(text_like(5000, 1), level=9)          # CDict row is a lazy strategy
    cd = zstd_amd.CDict(text_like(50000, 1), level=3)
    ctx = zstd_amd.Context(0, max_units=4, records_total_bytes=100000)
    with pytest.raises(zstd_amd.ZhipError):
        ctx.compress_records(cd, [text_like(140000, 2)])     # above 128 KB: not a single-block frame


@pytest.mark.parametrize("kind,level,dsize", [("text", 3, 110000), ("datagen", 3, 40000), ("text", 1, 110000), ("text", 4, 60000), ("text", -1, 20000), ("text", 3, 5)])
def test_copy_mode_sources_match_oracle_bytes(env, kind, level, dsize):
    """sources above the attach cut-off (8 KB for a strategy-fast CDict, 16 KB for dfast): the reference copies the dictionary's
    tables and runs its extDict block compressors; k_ext_init + k_parse_ext do the same, mixed with attach-mode records in one call"""
    lo, zstd_amd, torch = env
    rng = np.random.default_rng(abs(level) * 131 + dsize % 89)
    corpus = text_like(400000, 5) if kind == "text" else datagen(lo, 400000, 60, 5)
    dict_ = corpus[:dsize].copy()
    sizes = [8193, 9000, 16385, 20000, 50000, 100000, 131072, 500, 16384, 8192, 70000] + [int(x) for x in rng.integers(8193, 60000, size=40)]
    recs = records_of(corpus, rng, sizes)
    if dsize > 20000:
        recs.append(np.concatenate([dict_[-9000:], dict_[:9000], dict_[-40:]]))      # runs off the dictionary's end into the source
    recs.append(np.zeros(30000, np.uint8))
    recs.append(rng.integers(0, 256, size=20000, dtype=np.uint8))
    ctx = zstd_amd.Context(0, max_units=len(recs), records_total_bytes=sum(len(r) for r in recs))
    cd = zstd_amd.CDict(dict_, level=level)
    got, fs = ctx.compress_records(cd, recs, return_sizes=True)
    want = oracle_frames(lo, dict_, recs, level)
    pos = 0
    for i, (r, w) in enumerate(zip(recs, want)):
        g = got[pos:pos + int(fs[i])]
        pos += int(fs[i])
        assert g == w, (kind, level, dsize, i, len(r), len(g), len(w))
    assert pos == len(got)
    dd = zstd_amd.DDict(dict_)
    assert zstd_amd.DContext(0).decompress(got, ddict=dd) == b"".join(r.tobytes() for r in recs)


def test_many_records_use_the_tiled_prefix_sum(env):
    """more than 65 536 records in one call: the packed stream's offsets come from the tiled prefix sum (k_offsets_tiles / _apply) —
    it must equal the stream of the same records compressed in two calls of at most 65 536 (single-workgroup scan), and decode back"""
    import os, sys
    lo, zstd_amd, torch = env
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from zstd_amd import workloads as W
    zd = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "github_like_110k.zdict"), dtype=np.uint8)
    flat, offs = W.github_like_records_native(70000, seed=3)
    recs = [flat[int(offs[i]):int(offs[i]) + 150 + (i % 200)] for i in range(70000)]        # short records, ragged lengths
    tot = sum(len(r) for r in recs)
    cd = zstd_amd.CDict(zd, level=3)
    ctx = zstd_amd.Context(0, max_units=70000, records_total_bytes=tot)
    whole, fs = ctx.compress_records(cd, recs, return_sizes=True)
    a = ctx.compress_records(cd, recs[:40000])
    b = ctx.compress_records(cd, recs[40000:])
    assert whole == a + b
    assert int(np.asarray(fs).sum()) == len(whole)
    dd = zstd_amd.DDict(zd, device=0)
    back = zstd_amd.DContext(0).decompress(whole, ddict=dd, capacity=tot)
    assert back == b"".join(r.tobytes() for r in recs)
    ctx.close()
