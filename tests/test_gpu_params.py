"""B2 with EXPLICIT compression parameters: zhip_compress_params / ZSTD_CCtx_setParameter(ZSTD_c_windowLog ...)
through the shim must give the bytes the reference gives with the same parameters set on its CCtx
(lib/compress/zstd_compress.c:710-768, :1617-1644) — checked against the real reference where oracle/_ref travels, and against
committed golden digests made from it (tests/golden/params_v1.json)."""
import ctypes as C
import hashlib
import json
import os
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, datagen, text_like, _buf, ERR, ROOT

pytestmark = pytest.mark.gpu
UNIT = 131072

# (level, [windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy]); 0 = the level's own
PARAM_SETS = [
    (1, [19, 13, 14, 1, 7, 0, 1]),      # the reference's default level-1 row (srcSize unknown / > 256 KB) applied to 128 KB units
    (1, [0, 0, 12, 0, 0, 0, 0]),        # smaller table
    (1, [0, 0, 15, 0, 4, 0, 0]),        # largest LDS table, 4-byte hash
    (1, [0, 0, 0, 0, 5, 3, 0]),         # acceleration through targetLength (literals stay raw, internal.h:621-634)
    (1, [0, 0, 0, 0, 7, 0, 0]),
    (3, [0, 14, 15, 0, 6, 0, 0]),       # dfast with other table sizes / hash width
    (3, [0, 0, 14, 0, 0, 0, 1]),        # level 3's row run as ZSTD_fast (its hashLog 16 would not fit LDS: that case is in the shim test)
    (1, [0, 15, 16, 0, 5, 0, 2]),       # level 1 turned into dfast
    (5, [0, 15, 16, 4, 4, 0, 0]),       # greedy, hash chain: deeper search (reference run with the row matcher disabled)
    (6, [0, 0, 0, 2, 5, 0, 4]),         # lazy
]


def inputs(lo):
    return [("datagen", datagen(lo, 3 * UNIT + 4321, 50, 21)), ("text", text_like(2 * UNIT + 999, 5))]


def gpu_compress(zstd_amd, ctx, a, level, cp):
    L = zstd_amd.lib()
    L.zhip_compress_params.restype = C.c_size_t
    L.zhip_compress_params.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    cap = L.zhip_compressBound(len(a), UNIT)
    dst = np.zeros(cap, dtype=np.uint8)
    arr = (C.c_uint * 7)(*cp)
    r = L.zhip_compress_params(ctx._h, _buf(dst), cap, _buf(a), len(a), level, arr, UNIT, None)
    return r, dst


def test_explicit_parameters_match_reference_bytes():
    import zstd_amd
    lo = load_oracle()
    ctx = zstd_amd.Context(0, max_units=16)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "params_v1.json")))
    lr = load_ref() if have_ref() else None
    if lr is not None:
        lr.zref_compress_chunks_level_params.restype = C.c_size_t
        lr.zref_compress_chunks_level_params.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    for name, a in inputs(lo):
        for level, cp in PARAM_SETS:
            r, dst = gpu_compress(zstd_amd, ctx, a, level, cp)
            assert not zstd_amd.lib().zhip_isError(r), (name, level, cp, zstd_amd.lib().zhip_last_error(ctx._h))
            key = f"{name}|{level}|{','.join(map(str, cp))}"
            assert hashlib.sha256(dst[:r].tobytes()).hexdigest() == gold[key], ("golden digest", key)
            if lr is not None:
                want = np.zeros(len(dst) + 1024, dtype=np.uint8)
                arr = (C.c_int * 7)(*cp)
                k = lr.zref_compress_chunks_level_params(level, arr, 1 if cp[6] in (3, 4, 5) or (cp[6] == 0 and level >= 5) else 0, UNIT, _buf(a), len(a), _buf(want), len(want))
                assert k != ERR and k == r and want[:k].tobytes() == dst[:r].tobytes(), ("reference bytes", key)


def test_shim_honours_advanced_parameters_and_bounds():
    import zstd_amd
    from zstd_amd import build as zbuild
    lo = load_oracle()
    S = C.CDLL(zbuild.SHIM)
    S.ZSTD_createCCtx.restype = C.c_void_p
    S.ZSTD_CCtx_setParameter.restype = C.c_size_t; S.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    S.ZSTD_compress2.restype = C.c_size_t; S.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    S.ZSTD_compressCCtx.restype = C.c_size_t; S.ZSTD_compressCCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    S.ZSTD_compressBound.restype = C.c_size_t; S.ZSTD_compressBound.argtypes = [C.c_size_t]
    S.ZSTD_isError.restype = C.c_uint; S.ZSTD_isError.argtypes = [C.c_size_t]
    S.ZSTD_getErrorCode.restype = C.c_int; S.ZSTD_getErrorCode.argtypes = [C.c_size_t]
    S.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    ZSTD_c = {"level": 100, "windowLog": 101, "hashLog": 102, "chainLog": 103, "searchLog": 104, "minMatch": 105, "targetLength": 106, "strategy": 107}
    a = datagen(lo, UNIT, 50, 33)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "params_v1.json")))
    c = S.ZSTD_createCCtx()
    assert S.ZSTD_getErrorCode(S.ZSTD_CCtx_setParameter(c, ZSTD_c["windowLog"], 9)) == 42          # parameter_outOfBound
    assert S.ZSTD_getErrorCode(S.ZSTD_CCtx_setParameter(c, ZSTD_c["minMatch"], 8)) == 42
    S.ZSTD_CCtx_setParameter(c, ZSTD_c["level"], 1)
    for k, v in (("windowLog", 19), ("chainLog", 13), ("hashLog", 14), ("searchLog", 1), ("minMatch", 7), ("strategy", 1)):
        assert S.ZSTD_CCtx_setParameter(c, ZSTD_c[k], v) == 0
    cap = S.ZSTD_compressBound(len(a)); dst = np.zeros(cap, dtype=np.uint8)
    r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), len(a))
    assert not S.ZSTD_isError(r) and hashlib.sha256(dst[:r].tobytes()).hexdigest() == gold["shim|datagen33|1|19,13,14,1,7,0,1"]
    # ZSTD_compressCCtx ignores the advanced parameters (zstd_compress.c:5428): plain level 1
    r2 = S.ZSTD_compressCCtx(c, _buf(dst), cap, _buf(a), len(a), 1)
    assert not S.ZSTD_isError(r2) and hashlib.sha256(dst[:r2].tobytes()).hexdigest() == gold["shim|datagen33|1|plain"]
    # what the device cannot run says so: binary-tree strategies, a window smaller than the unit
    for k, v in (("strategy", 7), ("windowLog", 12)):
        c2 = S.ZSTD_createCCtx(); S.ZSTD_CCtx_setParameter(c2, ZSTD_c["level"], 1); S.ZSTD_CCtx_setParameter(c2, ZSTD_c[k], v)
        rr = S.ZSTD_compress2(c2, _buf(dst), cap, _buf(a), len(a))
        assert S.ZSTD_getErrorCode(rr) == 40, (k, v, S.ZSTD_getErrorCode(rr))
        S.ZSTD_freeCCtx(c2)
    # a ZSTD_fast table above the unit kernel's LDS bound (hashLog 16 .. 18; round 2 refused it) goes through the frame kernel: the reference's bytes
    lr = load_ref() if have_ref() else None
    for hl in (16, 18):
        c2 = S.ZSTD_createCCtx(); S.ZSTD_CCtx_setParameter(c2, ZSTD_c["level"], 1)
        S.ZSTD_CCtx_setParameter(c2, ZSTD_c["windowLog"], 20); S.ZSTD_CCtx_setParameter(c2, ZSTD_c["hashLog"], hl)
        for n in (len(a), 50000):
            rr = S.ZSTD_compress2(c2, _buf(dst), cap, _buf(a), n)
            assert not S.ZSTD_isError(rr), (hl, n, S.ZSTD_getErrorCode(rr))
            if lr is not None:
                want = np.zeros(cap + 1024, dtype=np.uint8)
                arr = (C.c_int * 7)(20, 0, hl, 0, 0, 0, 0)
                k = lr.zref_compress_chunks_level_params(1, arr, 0, UNIT, _buf(a), n, _buf(want), len(want))
                assert k != ERR and k == rr and want[:k].tobytes() == dst[:rr].tobytes(), ("reference bytes", hl, n)
        S.ZSTD_freeCCtx(c2)
    S.ZSTD_freeCCtx(c)
