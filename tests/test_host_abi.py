"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol include/zstd_hip.h
declares, parameter selection and the input generator agree with the oracle (and with the real reference when
oracle/_ref is present).  No compute call is made (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, _buf, ROOT, ERR


@pytest.fixture(scope="module")
def zlib_():
    from zstd_amd import build as zb
    import zstd_amd
    zb.build()
    return zstd_amd


def test_every_declared_symbol_is_exported(zlib_):
    hdr = open(os.path.join(ROOT, "include", "zstd_hip.h")).read()
    names = sorted(set(re.findall(r"\b(zhip_[a-zA-Z_]+)\s*\(", hdr)))
    assert len(names) >= 15
    L = C.CDLL(zlib_.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_dropin_shim_exports_every_declared_symbol(zlib_):
    """libzstd_hipshim.so: the ZSTD_* names of include/zstd_hip_dropin.h, all present, all backed by libzstd_hip only"""
    from zstd_amd import build as zb
    hdr = open(os.path.join(ROOT, "include", "zstd_hip_dropin.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)                      # declarations only, not the prose
    names = sorted(set(re.findall(r"\b(ZSTD_[a-zA-Z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 12 and "ZSTD_compress2" in names
    L = C.CDLL(zb.SHIM)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    out = subprocess.check_output(["ldd", zb.SHIM]).decode()
    assert "libzstd_hip.so" in out and "zoracle" not in out and "zstd_ref" not in out
    # host-only entry points behave like the reference's
    L.ZSTD_compressBound.restype = C.c_size_t
    L.ZSTD_compressBound.argtypes = [C.c_size_t]
    L.ZSTD_isError.argtypes = [C.c_size_t]
    assert L.ZSTD_compressBound(131072) == 131072 + 512 and L.ZSTD_isError(C.c_size_t(-40).value) == 1 and L.ZSTD_isError(1000) == 0
    assert L.ZSTD_defaultCLevel() == 3


def test_dropin_header_cites_the_reference_lines_it_stands_in_for():
    """include/zstd_hip_dropin.h declares the reference's prototypes again and names, per function, the line of lib/zstd.h it stands in for: the line
    cited holds that function's declaration (checked where the reference is on disk)"""
    import re
    ref_h = "/root/reference/lib/zstd.h"
    if not os.path.exists(ref_h):
        pytest.skip("reference sources not on this box")
    ref = open(ref_h).read().split("\n")
    seen = 0
    for l in open(os.path.join(ROOT, "include", "zstd_hip_dropin.h")).read().split("\n"):
        m = re.match(r"\s*[\w\s\*]+?\b(ZSTD_\w+)\s*\(.*\);\s*/\*\s*(?:lib/zstd\.h)?:(\d+)", l)
        if not m:
            continue
        name, cited = m.group(1), int(m.group(2))
        around = " ".join(ref[cited - 1: cited + 1])                 # the return type may sit on the line before the name
        assert re.search(r"\b" + name + r"\s*\(", around), (name, cited, ref[cited - 1])
        seen += 1
    assert seen >= 35


def test_no_oracle_or_reference_in_the_product_binary(zlib_):
    from zstd_amd import build as zb
    for lib in (zlib_.LIB_PATH, zb.SHIM):
        out = subprocess.check_output(["ldd", lib]).decode()
        assert "zoracle" not in out and "zstd_ref" not in out and "zref" not in out
    syms = subprocess.check_output(["nm", "-D", zlib_.LIB_PATH]).decode()
    assert " zo_" not in syms and "ZSTD_compress" not in syms


def test_cparams_match_oracle_and_reference(zlib_):
    lo = load_oracle()
    lr = load_ref() if have_ref() else None
    sizes = [0, 1, 5, 63, 64, 65, 255, 256, 1000, 4096, 16384, 16385, 65536, 131071, 131072, 131073, 262144, 262145, 1 << 20]
    for level in (-7, -1, 0, 1, 2, 3, 4, 5, 19):
        for n in sizes:
            mine = (C.c_uint * 7)()
            rc = zlib_.lib().zhip_getCParams(level, n, mine)
            o = (C.c_uint * 7)()
            ro = lo.zo_get_cparams(level, n, o)
            assert (rc == 0) == (ro == 0)
            if rc == 0:
                assert list(mine) == list(o), (level, n)
                if lr is not None and n > 0:
                    r = (C.c_int * 7)()
                    lr.zref_get_cparams(level, n, 0, r)
                    assert list(mine) == list(r), (level, n)


def test_datagen_both_modes(zlib_):
    lo = load_oracle()
    for P in (0, 35, 50, 100):
        a = zlib_.datagen(300001, P, seed=3, stream_mode=False)
        b = np.zeros(300001, dtype=np.uint8)
        lo.zo_datagen(_buf(b), 300001, P / 100.0, 0.0, 3)
        assert a.tobytes() == b.tobytes()
    exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
    if os.path.exists(exe):
        for n in (1, 131072, 500000):
            want = subprocess.check_output([exe, "stream", str(n), "50", "2"])
            assert zlib_.datagen(n, 50, seed=2, stream_mode=True).tobytes() == want


def test_compress_bound_matches(zlib_):
    lo = load_oracle()
    for n in (0, 1, 1000, 131072, 131073, 1 << 20, (1 << 20) + 5):
        want = 0
        off = 0
        while True:
            ln = min(131072, n - off)
            want += lo.zo_compress_bound(ln)
            off += ln
            if off >= n:
                break
        assert zlib_.compress_bound(n, 131072) == want


def test_calls_fail_loudly_without_a_gpu(zlib_):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zlib_.ZhipError):
        zlib_.Context(0, 4)


def test_seek_table_bytes_match_the_reference_writer(zlib_):
    """zhip_write_seek_table == ZSTD_seekable_writeSeekTable (contrib/seekable_format/zstdseek_compress.c) on the same frame log"""
    if not have_ref():
        pytest.skip("reference build (oracle/_ref) not present")
    lr = load_ref()
    lr.zref_seek_table.restype = C.c_size_t
    lr.zref_seek_table.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    L = zlib_.lib()
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 17, 1000):
        cs = rng.integers(10, 140000, size=max(n, 1), dtype=np.uint32)
        ds = rng.integers(0, 131073, size=max(n, 1), dtype=np.uint32)
        ck = rng.integers(0, 2**32, size=max(n, 1), dtype=np.uint64).astype(np.uint32)
        for with_ck in (False, True):
            cap = L.zhip_seek_table_bound(n, 1)
            mine = np.zeros(cap, dtype=np.uint8)
            ref = np.zeros(cap + 64, dtype=np.uint8)
            r = L.zhip_write_seek_table(_buf(mine), cap, _buf(cs), _buf(ds), _buf(ck) if with_ck else None, n)
            k = lr.zref_seek_table(_buf(ref), len(ref), _buf(cs), _buf(ds), _buf(ck) if with_ck else None, n)
            assert k != ERR and r == k == L.zhip_seek_table_bound(n, int(with_ck))
            assert mine[:r].tobytes() == ref[:k].tobytes(), (n, with_ck)


def test_frame_walker_matches_the_oracle(zlib_):
    """zhip_find_frames / zhip_frame_compressed_size (host code of the decoder) against oracle/zoracle_dec.c's zo_frame_info"""
    lo = load_oracle()
    from _libs import datagen, text_like
    parts = [datagen(lo, 300000, 50, 1), text_like(5000, 2), np.zeros(0, dtype=np.uint8), text_like(131072, 3)]
    stream = b""
    want = []
    for i, a in enumerate(parts):
        n = len(a)
        cap = lo.zo_compress_bound(131072) * (n // 131072 + 1) + 64
        dst = np.empty(cap, dtype=np.uint8)
        r = lo.zo_compress_chunks(1 + 2 * (i & 1), 131072, _buf(a) if n else None, n, _buf(dst), cap, None, 0)
        assert r != ERR
        pos = 0
        while pos < r:
            cs, ds = C.c_size_t(0), C.c_ulonglong(0)
            assert lo.zo_frame_info(_buf(dst[pos:r]), r - pos, C.byref(cs), C.byref(ds)) == 0
            want.append((len(stream) + pos, cs.value, ds.value)); pos += cs.value
        stream += dst[:r].tobytes()
        if i == 1:
            stream += b"\x50\x2a\x4d\x18\x04\x00\x00\x00skip"           # a skippable frame between two frames
    fr = zlib_.find_frames(stream)
    got = list(zip(fr["src_off"].tolist(), fr["src_size"].tolist(), fr["content"].tolist()))
    assert got == want and (fr["bound"] >= fr["content"]).all()
    L = zlib_.lib()
    L.zhip_frame_compressed_size.restype = C.c_size_t
    L.zhip_frame_compressed_size.argtypes = [C.c_void_p, C.c_size_t]
    b = np.frombuffer(stream, dtype=np.uint8)
    assert L.zhip_frame_compressed_size(_buf(b), len(b)) == want[0][1]
    with pytest.raises(zlib_.ZhipError):
        zlib_.find_frames(stream + b"\x00")                               # trailing garbage: srcSize_wrong
    with pytest.raises(zlib_.ZhipError):
        zlib_.find_frames(stream[:-3])                                     # truncated last frame


def test_shim_frame_inspection_matches_the_reference(zlib_):
    """host-only ZSTD_* inspectors of the shim (no GPU needed) against the real reference on its own frames"""
    if not have_ref():
        pytest.skip("needs oracle/_ref")
    from zstd_amd import build as zb
    from _libs import text_like
    lr = load_ref(); lo = load_oracle()
    S = C.CDLL(zb.SHIM)
    R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so"))
    for L in (S, R):
        L.ZSTD_decompressBound.restype = C.c_ulonglong; L.ZSTD_decompressBound.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_findDecompressedSize.restype = C.c_ulonglong; L.ZSTD_findDecompressedSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_getFrameContentSize.restype = C.c_ulonglong; L.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_findFrameCompressedSize.restype = C.c_size_t; L.ZSTD_findFrameCompressedSize.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_isFrame.restype = C.c_uint; L.ZSTD_isFrame.argtypes = [C.c_void_p, C.c_size_t]
        L.ZSTD_getDictID_fromFrame.restype = C.c_uint; L.ZSTD_getDictID_fromFrame.argtypes = [C.c_void_p, C.c_size_t]
    a = text_like(300000, 7)
    streams = []
    for cs, ck, wl in ((1, 0, 0), (0, 1, 0), (0, 0, 12)):
        cap = int(lr.zref_compress_bound(len(a))) + 64
        dst = np.empty(cap, dtype=np.uint8)
        r = lr.zref_compress_frame_params(3, cs, ck, wl, _buf(a), len(a), _buf(dst), cap)
        assert r != ERR
        streams.append(dst[:r].tobytes())
    streams.append(streams[0] + b"\x52\x2a\x4d\x18\x02\x00\x00\x00zz" + streams[1])
    zd = np.fromfile(os.path.join(ROOT, "tests", "golden", "github_like_110k.zdict"), dtype=np.uint8)
    sizes = (C.c_size_t * 1)(2000); outs = (C.c_size_t * 1)()
    dst = np.empty(4096, dtype=np.uint8)
    assert lr.zref_compress_records_cdict(3, _buf(zd), len(zd), _buf(a), sizes, 1, _buf(dst), 4096, outs) != ERR
    streams.append(dst[:outs[0]].tobytes())
    for z in streams:
        b = np.frombuffer(z + b"\x00" * 8, dtype=np.uint8); n = len(z)
        for fn in ("ZSTD_decompressBound", "ZSTD_findDecompressedSize", "ZSTD_getFrameContentSize", "ZSTD_findFrameCompressedSize", "ZSTD_isFrame", "ZSTD_getDictID_fromFrame"):
            assert getattr(S, fn)(_buf(b), n) == getattr(R, fn)(_buf(b), n), fn
