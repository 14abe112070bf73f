"""-m gpu: the ZSTD_*-named drop-in (libzstd_hipshim.so, include/zstd_hip_dropin.h = boundary B2 of SURVEY.md 8b).
A caller written against zstd.h: createCCtx / setParameter / compress2 / compressCCtx / compress.  Checked against
the REAL reference when oracle/_ref is present (bytes for <= 128 KB inputs, ZSTD_decompress round trip for larger
ones), else against the oracle restatement."""
import ctypes as C
import os
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, corpus_cases, datagen, _buf, ROOT, ERR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import zstd_amd                                    # loads torch's HIP runtime first, then libzstd_hip
    zstd_amd.lib()
    from zstd_amd import build as zb
    S = C.CDLL(zb.SHIM)
    S.ZSTD_createCCtx.restype = C.c_void_p
    S.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    S.ZSTD_CCtx_setParameter.restype = C.c_size_t
    S.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    S.ZSTD_CCtx_reset.restype = C.c_size_t
    S.ZSTD_CCtx_reset.argtypes = [C.c_void_p, C.c_int]
    for f in ("ZSTD_compress2",):
        getattr(S, f).restype = C.c_size_t
        getattr(S, f).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    S.ZSTD_compressCCtx.restype = C.c_size_t
    S.ZSTD_compressCCtx.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    S.ZSTD_compress.restype = C.c_size_t
    S.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    S.ZSTD_compressBound.restype = C.c_size_t
    S.ZSTD_compressBound.argtypes = [C.c_size_t]
    S.ZSTD_isError.argtypes = [C.c_size_t]
    S.ZSTD_getErrorName.restype = C.c_char_p
    S.ZSTD_getErrorName.argtypes = [C.c_size_t]
    return S, load_oracle(), (load_ref() if have_ref() else None)


def expect_unit(lo, lr, a, level):
    """the reference's ZSTD_compress2 of one <= 128 KB input (real library if present, else the oracle)"""
    cap = lo.zo_compress_bound(len(a)) + 64
    dst = np.zeros(cap, dtype=np.uint8)
    if lr is not None:
        r = lr.zref_compress_frame(level, _buf(a), len(a), _buf(dst), cap)
    else:
        r = lo.zo_compress_unit(_buf(dst), cap, _buf(a), len(a), level)
    assert r != ERR
    return dst[:r].tobytes()


def shim_compress2(S, a, level=None, cap=None):
    c = S.ZSTD_createCCtx()
    if level is not None:
        assert S.ZSTD_CCtx_setParameter(c, 100, level) == 0
    cap = S.ZSTD_compressBound(len(a)) if cap is None else cap
    dst = np.zeros(max(cap, 1), dtype=np.uint8)
    r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a) if len(a) else None, len(a))
    S.ZSTD_freeCCtx(c)
    return r, dst


@pytest.mark.parametrize("level", [1, 3, None, -5])
def test_single_unit_is_byte_identical_to_the_reference(env, level):
    S, lo, lr = env
    for n in (0, 5, 100, 4096, 70000, 131072):
        for name, a in list(corpus_cases(lo, sizes=(n,), seeds=(9,)))[:6]:
            r, dst = shim_compress2(S, a, level)
            assert not S.ZSTD_isError(r), (name, S.ZSTD_getErrorName(r))
            assert dst[:r].tobytes() == expect_unit(lo, lr, a, 3 if level is None else level), (name, level)


def test_large_input_is_a_valid_stream_of_unit_frames(env):
    S, lo, lr = env
    n = 5 * 131072 + 4321
    a = datagen(lo, n, 50, 11)
    r, dst = shim_compress2(S, a, 1)
    assert not S.ZSTD_isError(r)
    # equals the reference run per 128 KB chunk (zstd -b1 -B128K) ...
    cap = lo.zo_compress_bound(131072) * 6
    want = np.zeros(cap, dtype=np.uint8)
    w = lo.zo_compress_chunks(1, 131072, _buf(a), n, _buf(want), cap, None, 0)
    assert dst[:r].tobytes() == want[:w].tobytes()
    # ... and the reference decoder restores the input from the concatenated frames
    if lr is not None:
        out = np.zeros(n, dtype=np.uint8)
        d = lr.zref_decompress(_buf(out), n, _buf(dst), r)
        assert d == n and out.tobytes() == a.tobytes()


def test_compressCCtx_ignores_cctx_level_and_one_shot_works(env):
    S, lo, lr = env
    a = datagen(lo, 50000, 60, 12)
    c = S.ZSTD_createCCtx()
    assert S.ZSTD_CCtx_setParameter(c, 100, 3) == 0
    cap = S.ZSTD_compressBound(len(a))
    dst = np.zeros(cap, dtype=np.uint8)
    r = S.ZSTD_compressCCtx(c, _buf(dst), cap, _buf(a), len(a), 1)
    assert dst[:r].tobytes() == expect_unit(lo, lr, a, 1)
    r2 = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), len(a))         # the cctx still holds level 3
    assert dst[:r2].tobytes() == expect_unit(lo, lr, a, 3)
    S.ZSTD_freeCCtx(c)
    r3 = S.ZSTD_compress(_buf(dst), cap, _buf(a), len(a), 2)
    assert dst[:r3].tobytes() == expect_unit(lo, lr, a, 2)


def test_errors_use_the_reference_codes_and_nothing_falls_back_to_the_host(env):
    S, lo, lr = env
    a = datagen(lo, 20000, 50, 13)
    r, _ = shim_compress2(S, a, 19)                                    # btultra2: not on the device, no CPU fallback
    assert S.ZSTD_isError(r) and r == C.c_size_t(-40).value            # parameter_unsupported
    c = S.ZSTD_createCCtx()
    assert S.ZSTD_CCtx_setParameter(c, 201, 1) == 0                    # checksumFlag: XXH64 on the device
    if lr is not None:
        lr.zref_compress_chunks_checksum.restype = C.c_size_t
        lr.zref_compress_chunks_checksum.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        assert S.ZSTD_CCtx_setParameter(c, 100, 3) == 0
        cap = S.ZSTD_compressBound(len(a)); d1 = np.zeros(cap, dtype=np.uint8); d2 = np.zeros(cap + 64, dtype=np.uint8)
        k1 = S.ZSTD_compress2(c, _buf(d1), cap, _buf(a), len(a))
        k2 = lr.zref_compress_chunks_checksum(3, 131072, _buf(a), len(a), _buf(d2), len(d2), None, 0)
        assert not S.ZSTD_isError(k1) and d1[:k1].tobytes() == d2[:k2].tobytes()
        k3 = S.ZSTD_compressCCtx(c, _buf(d1), cap, _buf(a), len(a), 3)          # ignores the flag, like the reference
        assert d1[:k3].tobytes() == expect_unit(lo, lr, a, 3)
    assert S.ZSTD_isError(S.ZSTD_CCtx_setParameter(c, 160, 1))         # ZSTD_c_enableLongDistanceMatching: not on the device
    assert S.ZSTD_CCtx_setParameter(c, 400, 2) == 0 and S.ZSTD_isError(S.ZSTD_CCtx_setParameter(c, 400, 257))   # nbWorkers: accepted (job-pool frames), bounded
    assert S.ZSTD_CCtx_setParameter(c, 400, 0) == 0
    assert S.ZSTD_CCtx_setParameter(c, 101, 0) == 0                    # windowLog 0 = default
    S.ZSTD_freeCCtx(c)
    r, dst = shim_compress2(S, a, 1, cap=64)                           # far too small
    assert S.ZSTD_isError(r) and r == C.c_size_t(-70).value            # dstSize_tooSmall
    want = expect_unit(lo, lr, a, 1)
    r, dst = shim_compress2(S, a, 1, cap=len(want))                    # exactly enough, below ZSTD_compressBound
    assert not S.ZSTD_isError(r) and dst[:r].tobytes() == want


def test_cdict_entry_points_match_the_reference(env):
    """ZSTD_createCDict / ZSTD_CCtx_refCDict / ZSTD_compress2 / ZSTD_compress_usingCDict of the shim vs the real library
    (or the oracle where oracle/_ref is absent): trained ZDICT fixture, JSON records"""
    import sys
    S, lo, lr = env
    sys.path.insert(0, ROOT)
    from zstd_amd import workloads as W
    S.ZSTD_createCDict.restype = C.c_void_p
    S.ZSTD_createCDict.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    S.ZSTD_freeCDict.argtypes = [C.c_void_p]
    S.ZSTD_CCtx_refCDict.restype = C.c_size_t
    S.ZSTD_CCtx_refCDict.argtypes = [C.c_void_p, C.c_void_p]
    S.ZSTD_compress_usingCDict.restype = C.c_size_t
    S.ZSTD_compress_usingCDict.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    zd = np.fromfile(os.path.join(ROOT, "tests", "golden", "github_like_110k.zdict"), dtype=np.uint8)
    flat, offs = W.github_like_records(40, seed=11)
    recs = [flat[int(offs[i]):int(offs[i + 1])].copy() for i in range(40)]
    lo.zo_cdict_create.restype = C.c_void_p
    lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lo.zo_compress_unit_cdict.restype = C.c_size_t
    lo.zo_compress_unit_cdict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    ocd = lo.zo_cdict_create(_buf(zd), len(zd), 3)
    cd = S.ZSTD_createCDict(_buf(zd), len(zd), 3)
    assert cd
    c = S.ZSTD_createCCtx()
    assert S.ZSTD_CCtx_refCDict(c, cd) == 0
    for i, r in enumerate(recs):
        cap = S.ZSTD_compressBound(len(r))
        dst = np.zeros(cap, dtype=np.uint8)
        k = S.ZSTD_compress2(c, _buf(dst), cap, _buf(r), len(r)) if i % 2 == 0 else S.ZSTD_compress_usingCDict(c, _buf(dst), cap, _buf(r), len(r), cd)
        assert not S.ZSTD_isError(k), S.ZSTD_getErrorName(k)
        want = np.zeros(len(r) + 700, dtype=np.uint8)
        w = lo.zo_compress_unit_cdict(_buf(want), len(want), _buf(r), len(r), ocd)
        assert dst[:k].tobytes() == want[:w].tobytes(), i
    big = np.concatenate(recs)[:30000]
    dst = np.zeros(S.ZSTD_compressBound(len(big)), dtype=np.uint8)
    k = S.ZSTD_compress2(c, _buf(dst), len(dst), _buf(big), len(big))       # above the attach cut-off: the dictionary's copy mode
    assert not S.ZSTD_isError(k), S.ZSTD_getErrorName(k)
    want = np.zeros(len(big) + 700, dtype=np.uint8)
    w = lo.zo_compress_unit_cdict(_buf(want), len(want), _buf(big), len(big), ocd)
    assert dst[:k].tobytes() == want[:w].tobytes()
    huge = np.concatenate([np.concatenate(recs)] * 4)[:140000]
    dst2 = np.zeros(S.ZSTD_compressBound(len(huge)), dtype=np.uint8)
    k = S.ZSTD_compress2(c, _buf(dst2), len(dst2), _buf(huge), len(huge))   # above 128 KB with a dictionary: not a single-block frame
    assert S.ZSTD_isError(k) and b"Unsupported" in S.ZSTD_getErrorName(k)
    assert S.ZSTD_CCtx_reset(c, 3) == 0                                     # parameters reset: the dictionary is dropped
    k = S.ZSTD_compress2(c, _buf(dst), len(dst), _buf(big), len(big))
    assert not S.ZSTD_isError(k)
    S.ZSTD_freeCCtx(c); S.ZSTD_freeCDict(cd)


def test_c_program_against_the_shim(tmp_path):
    """examples/roundtrip.c: plain C against the reference's names, linked with -lzstd_hipshim; compiled here with gcc (with the
    reference's own zstd.h when /root/reference exists, else include/zstd_hip_dropin.h) and run on the GPU"""
    import subprocess, shutil
    import zstd_amd
    from _libs import ROOT
    libdir = os.path.dirname(zstd_amd.LIB_PATH)
    exe = str(tmp_path / "roundtrip")
    ref_hdr = "/root/reference/lib/zstd.h"
    cmd = ["gcc", "-O2", "-Wall", "-Werror"]
    cmd += ["-DUSE_REFERENCE_HEADER", "-I/root/reference/lib"] if os.path.exists(ref_hdr) else ["-I" + os.path.join(ROOT, "include")]
    cmd += [os.path.join(ROOT, "examples", "roundtrip.c"), "-L" + libdir, "-lzstd_hipshim", "-lzstd_hip", "-Wl,-rpath," + libdir, "-o", exe]
    subprocess.check_call(cmd)
    for n, level in ((5 << 20, 3), (100, 1), (131072, 5), (0, 1), (3_000_000, 1)):
        out = subprocess.run([exe, str(n), str(level)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and "roundtrip ok" in out.stdout, (n, level, out.stdout, out.stderr)


def test_compressStream2_one_shot_equals_compress2(env):
    """ZSTD_compressStream2(ZSTD_e_end) with all input and ZSTD_compressBound of room is ZSTD_compress2 in the reference
    (zstd_compress.c:6069-6084) and here; the streaming state machine itself is not on the device: parameter_unsupported"""
    S, lo, lr = env
    class InB(C.Structure):
        _fields_ = [("src", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]
    class OutB(C.Structure):
        _fields_ = [("dst", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]
    S.ZSTD_compressStream2.restype = C.c_size_t
    S.ZSTD_compressStream2.argtypes = [C.c_void_p, C.POINTER(OutB), C.POINTER(InB), C.c_int]
    S.ZSTD_createCStream.restype = C.c_void_p
    S.ZSTD_freeCStream.argtypes = [C.c_void_p]
    S.ZSTD_initCStream.restype = C.c_size_t
    S.ZSTD_initCStream.argtypes = [C.c_void_p, C.c_int]
    a = datagen(lo, 100_000, 50, 3)
    zcs = S.ZSTD_createCStream()
    assert S.ZSTD_initCStream(zcs, 3) == 0
    cap = S.ZSTD_compressBound(len(a))
    dst = np.zeros(cap + 16, dtype=np.uint8)
    o = OutB(dst.ctypes.data, cap + 16, 16); i = InB(a.ctypes.data, len(a), 0)
    assert S.ZSTD_compressStream2(zcs, C.byref(o), C.byref(i), 2) == 0 and i.pos == len(a)
    r, want = shim_compress2(S, a, 3)
    assert o.pos - 16 == r and dst[16:o.pos].tobytes() == want[:r].tobytes()
    i = InB(a.ctypes.data, len(a), 0); o = OutB(dst.ctypes.data, cap, 0)
    assert S.ZSTD_isError(S.ZSTD_compressStream2(zcs, C.byref(o), C.byref(i), 0))          # ZSTD_e_continue: the state machine is the reference's
    o = OutB(dst.ctypes.data, 100, 0)
    assert S.ZSTD_isError(S.ZSTD_compressStream2(zcs, C.byref(o), C.byref(i), 2))          # too little room for the one-shot form
    S.ZSTD_freeCStream(zcs)


def test_row_matcher_parameter_is_not_sticky(env, monkeypatch):
    """ZSTD_c_useRowMatchFinder = disable, compress, ZSTD_CCtx_reset(parameters), compress again — the second frame must be
    the reference's DEFAULT (row-hash matcher) frame again, the first its hash-chain frame; level 5, one 128 KB unit each."""
    S, lo, lr = env
    if lr is None:
        pytest.skip("needs oracle/_ref")
    monkeypatch.delenv("ZHIP_ROW_MATCHER", raising=False)          # (conftest pins the hash chain for the older tests; the device context reads it at creation)
    lr.zref_compress_chunks_norow.restype = C.c_size_t
    lr.zref_compress_chunks_norow.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    a = datagen(lo, 131072, 50, 21)
    cap = S.ZSTD_compressBound(len(a))
    want_def = expect_unit(lo, lr, a, 5)
    d2 = np.zeros(cap + 64, dtype=np.uint8)
    r2 = lr.zref_compress_chunks_norow(5, 131072, _buf(a), len(a), _buf(d2), len(d2), None, 0)
    assert r2 != ERR
    want_hc = d2[:r2].tobytes()
    assert want_def != want_hc                          # the two matchers really differ on this input
    c = S.ZSTD_createCCtx()
    assert S.ZSTD_CCtx_setParameter(c, 100, 5) == 0
    assert S.ZSTD_CCtx_setParameter(c, 1011, 2) == 0   # ZSTD_c_useRowMatchFinder = ZSTD_ps_disable
    dst = np.zeros(cap, dtype=np.uint8)
    r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), len(a))
    assert not S.ZSTD_isError(r) and dst[:r].tobytes() == want_hc
    assert S.ZSTD_CCtx_reset(c, 2) == 0                 # ZSTD_reset_parameters
    assert S.ZSTD_CCtx_setParameter(c, 100, 5) == 0
    r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), len(a))
    assert not S.ZSTD_isError(r) and dst[:r].tobytes() == want_def
    assert S.ZSTD_CCtx_setParameter(c, 1011, 2) == 0
    assert S.ZSTD_CCtx_setParameter(c, 1011, 0) == 0   # explicit ZSTD_ps_auto after disable
    r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), len(a))
    assert not S.ZSTD_isError(r) and dst[:r].tobytes() == want_def
    # ZSTD_ps_enable on a source the reference gives windowLog <= 14: refused, not compressed differently
    small = a[:9000].copy()
    assert S.ZSTD_CCtx_setParameter(c, 1011, 1) == 0
    r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(small), len(small))
    assert S.ZSTD_isError(r)
    S.ZSTD_freeCCtx(c)


def test_a_source_of_256_MiB_takes_the_lanes_and_keeps_its_bytes(env):
    """From 256 MiB on (SHIM_LANES_MIN) ZSTD_compress2 through the shim runs on the lanes of zhip_compress_multi on the CCtx's one device (overlapped staging copies,
    PCIe transfers and kernels) instead of the plain one-stream call: the stream must be the one a single context makes, with a destination sized by ZSTD_compressBound
    alone, and a small source compressed by the SAME CCtx afterwards (plain path again) must still equal the reference's frame."""
    import hashlib
    import zstd_amd
    S, lo, lr = env
    n = (256 << 20) + 3 * 131072 + 777
    a = zstd_amd.datagen(n, 50, seed=5, stream_mode=True)
    c = S.ZSTD_createCCtx()
    assert S.ZSTD_CCtx_setParameter(c, 100, 1) == 0
    cap = S.ZSTD_compressBound(n)
    dst = np.zeros(cap, dtype=np.uint8)
    r = S.ZSTD_compress2(c, _buf(dst), cap, _buf(a), n)
    assert not S.ZSTD_isError(r), S.ZSTD_getErrorName(r)
    ctx = zstd_amd.Context(0, max_units=n // 131072 + 2)
    want = ctx.compress(a, level=1)
    ctx.close()
    assert r == len(want) and hashlib.sha256(dst[:r].tobytes()).digest() == hashlib.sha256(want).digest()
    small = datagen(lo, 100000, 40, 3)
    d2 = np.zeros(S.ZSTD_compressBound(len(small)), dtype=np.uint8)
    r2 = S.ZSTD_compress2(c, _buf(d2), len(d2), _buf(small), len(small))
    assert not S.ZSTD_isError(r2) and d2[:r2].tobytes() == expect_unit(lo, lr, small, 1)
    S.ZSTD_freeCCtx(c)
