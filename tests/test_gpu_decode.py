"""GPU parity of the decoder through the C ABI (zhip_decompress / zhip_decompress_frames_device / the ZSTD_* shim):
decoded bytes == the original input == what the oracle decoder (and, where oracle/_ref travelled, the real reference) returns."""
import base64, ctypes as C, hashlib, json, os, zlib
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, _buf, ERR, datagen, text_like, oracle_decompress, corpus_cases

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_v1.json")


def unpack(s):
    return zlib.decompress(base64.b64decode(s))


@pytest.fixture(scope="module")
def env():
    import torch
    import zstd_amd
    assert torch.cuda.is_available()
    return zstd_amd, load_oracle(), zstd_amd.DContext(0), torch


def test_golden_decode_vectors(env):
    z, lo, dctx, _ = env
    g = json.load(open(GOLD))
    vs = g["fixtures"] + g["frames"]
    for v in vs:
        out = dctx.decompress(unpack(v["zst"]), capacity=v["size"])
        assert len(out) == v["size"] and hashlib.sha256(out).hexdigest() == v["sha256"], v["name"]
    # all of them as ONE stream of concatenated frames (with a skippable frame in between)
    stream = b"".join(unpack(v["zst"]) for v in vs[:6]) + b"\x5a\x2a\x4d\x18\x03\x00\x00\x00abc" + b"".join(unpack(v["zst"]) for v in vs[6:])
    out = dctx.decompress(stream, capacity=sum(v["size"] for v in vs))
    assert hashlib.sha256(out).hexdigest() == hashlib.sha256(b"".join(oracle_decompress(lo, unpack(v["zst"]), v["size"] + 8) for v in vs)).hexdigest()
    for v in g["errors"]:
        with pytest.raises(z.ZhipError):
            dctx.decompress(unpack(v["zst"]), capacity=1 << 16)


def test_round_trip_of_the_product_frames(env):
    z, lo, dctx, _ = env
    ctx = z.Context(0, max_units=64)
    for level in (1, 3, 5, 7, -3):
        parts = [a for n in (1, 7, 1000, 70000, 131072) for _, a in corpus_cases(lo, sizes=(n,), seeds=(3,))]
        for a in parts[:: 3 if level not in (1, 3) else 1]:
            frames = ctx.compress(a, level=level)
            assert dctx.decompress(frames, capacity=len(a)) == a.tobytes(), (level, len(a))
    big = np.concatenate([datagen(lo, 3_000_000, 50, 4), text_like(2_000_000, 5), np.zeros(300_000, np.uint8), np.random.default_rng(1).integers(0, 256, 500_000, dtype=np.uint8)])
    for level in (1, 3):
        frames = ctx.compress(big, level=level)
        out = dctx.decompress(frames)
        assert out == big.tobytes()
    ctx.set_checksum(True)
    frames = ctx.compress(big[:1_000_000], level=1)
    assert dctx.decompress(frames) == big[:1_000_000].tobytes()
    bad = bytearray(frames); bad[-2] ^= 0x40                     # the last frame's checksum
    with pytest.raises(z.ZhipError, match="22"):
        dctx.decompress(bytes(bad))


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (travels to the GPU box as a binary)")
def test_reference_frames_and_corruptions(env):
    z, lo, dctx, _ = env
    lr = load_ref()
    from test_oracle_decode import ref_frame
    rng = np.random.default_rng(33)
    for kind, n, level in (("text", 3_000_000, 3), ("P50", 2_000_000, 1), ("text", 200_000, 19), ("P80", 1_400_000, -3), ("text", 90_000, 7),
                           ("P50", 600_000, 12), ("P20", 400_000, 16), ("text", 5, 3)):
        a = text_like(n, 5) if kind == "text" else datagen(lo, n, int(kind[1:]), 6)
        assert dctx.decompress(ref_frame(lr, a, level), capacity=n) == a.tobytes(), (kind, n, level)
    base = ref_frame(lr, text_like(20000, 9), 3)
    muts = []
    for _ in range(300):
        b = bytearray(base)
        k = rng.integers(0, 3)
        if k == 0:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:
            del b[int(rng.integers(9, len(b))):]
        muts.append(bytes(b))
    nerr = 0
    for m in muts:
        w = oracle_decompress(lo, m, 32768)
        try:
            got = dctx.decompress(m, capacity=32768)
        except z.ZhipError:
            got = None
        if w is None:
            assert got is None, m.hex()[:60]
            nerr += 1
        else:
            assert got == w
    assert nerr > 80
    # the same mutants (and more families) as ONE batch: good and bad frames interleaved on the same workgroups, per-frame status
    _, _, _, torch = env
    bases = [base, ref_frame(lr, datagen(lo, 30000, 50, 3), 1), ref_frame(lr, text_like(9000, 4), 19), ref_frame(lr, text_like(400, 6), 1)]
    batch = list(muts)
    for it in range(1500):
        b = bytearray(bases[it % len(bases)])
        k = rng.integers(0, 4)
        if k == 0:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif k == 2:
            del b[int(rng.integers(9, len(b))):]
        batch.append(bytes(b))                                  # k == 3: an intact frame between the broken ones
    cap = 32768
    blob = np.frombuffer(b"".join(batch) + b"\x00" * 64, dtype=np.uint8)
    src = torch.from_numpy(blob.copy()).cuda()
    out = torch.zeros(len(batch) * cap + 64, dtype=torch.uint8, device="cuda")
    ssz = np.array([len(m) for m in batch], dtype=np.uint64)
    so = np.concatenate([[0], np.cumsum(ssz)[:-1]]).astype(np.uint64)
    do = np.arange(len(batch), dtype=np.uint64) * cap
    r, status, dsz = dctx.decompress_frames_device(out.data_ptr(), do, np.full(len(batch), cap, np.uint64), src.data_ptr(), so, ssz, check=False)
    host = out.cpu().numpy()
    for i, m in enumerate(batch):
        w = oracle_decompress(lo, m, cap)
        if w is None:
            assert status[i] != 0, (i, m.hex()[:60])
        else:
            assert status[i] == 0 and host[i * cap: i * cap + int(dsz[i])].tobytes() == w, (i, int(status[i]))


def test_dictionary_round_trip(env):
    z, lo, dctx, _ = env
    zd = np.fromfile(os.path.join(os.path.dirname(GOLD), "github_like_110k.zdict"), dtype=np.uint8)
    rng = np.random.default_rng(12)
    for dict_ in (zd, text_like(30000, 41)):
        recs = []
        for i in range(500):
            n = int(rng.integers(20, 4000))
            st = int(rng.integers(0, len(dict_) - n))
            r = dict_[st:st + n].copy()
            r[rng.integers(0, n, size=max(1, n // 40))] = rng.integers(32, 127, size=max(1, n // 40), dtype=np.uint8)
            recs.append(r)
        ctx = z.Context(0, max_units=len(recs), records_total_bytes=sum(len(r) for r in recs))
        cd = z.CDict(dict_, level=3)
        frames = ctx.compress_records(cd, recs)
        dd = z.DDict(dict_)
        assert dctx.decompress(frames, ddict=dd) == b"".join(r.tobytes() for r in recs)
        if dict_ is zd:
            assert dd.dict_id != 0
            with pytest.raises(z.ZhipError, match="32"):         # dictionary_wrong: the frames name a dictionary we did not give
                dctx.decompress(frames)


def test_device_buffers_full_size(env):
    """256 MiB of datagen: compress on the device, decode on the device, compare on the device"""
    z, lo, dctx, torch = env
    n = 256 << 20
    host = z.datagen(n, 50, seed=3, stream_mode=True)
    src = torch.from_numpy(host).cuda()
    units = n // 131072
    ctx = z.Context(0, max_units=units)
    cap = z.compress_bound(n)
    comp = torch.empty(cap, dtype=torch.uint8, device="cuda")
    sizes = torch.empty(units, dtype=torch.int32, device="cuda")
    total = ctx.compress_device(comp.data_ptr(), cap, src.data_ptr(), n, level=1, sizes_ptr=sizes.data_ptr())
    csz = sizes.cpu().numpy().astype(np.uint64)
    assert int(csz.sum()) == total
    so = np.concatenate([[0], np.cumsum(csz)[:-1]]).astype(np.uint64)
    do = (np.arange(units, dtype=np.uint64) * 131072)
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    r, status, dsz = dctx.decompress_frames_device(out.data_ptr(), do, np.full(units, 131072, np.uint64), comp.data_ptr(), so, csz)
    assert r == n and not status.any() and (dsz == 131072).all()
    assert torch.equal(out, src)
    t = dctx.timing()
    print(f"decode {n >> 20} MiB: {t['decode_ms']:.2f} ms = {n / t['decode_ms'] / 1e6:.1f} GB/s")


def test_many_small_frames_every_workgroup_takes_several(env):
    """the decode kernel's frame queue: far more frames than resident workgroups, of mixed kinds (compressed / raw / rle / empty-ish
    tails) and sizes, so every workgroup goes round its take-a-frame loop many times and the frames finish out of order.  (Round 3's
    stall sat in exactly this loop: the leader's queue take and the workgroup barrier after it — DESIGN.md 4.6c.)"""
    z, lo, dctx, torch = env
    rng = np.random.default_rng(11)
    unit = 1024
    n = 12_000 * unit - 333
    parts = [text_like(n // 3, 7), datagen(lo, n // 3, 40, 9), np.zeros(40_000, np.uint8), rng.integers(0, 256, 60_000, dtype=np.uint8)]
    src = np.concatenate(parts + [text_like(n - sum(len(p) for p in parts), 8)])
    assert len(src) == n
    ctx = z.Context(0, max_units=n // unit + 1)
    for level in (1, 3):
        frames, sizes = ctx.compress(src, level=level, unit_size=unit, return_sizes=True)
        assert len(sizes) == n // unit + 1
        assert dctx.decompress(frames, capacity=n) == src.tobytes(), level
    # the same frames, device to device, in a shuffled order of destinations
    csz = np.asarray(sizes, dtype=np.uint64)
    so = np.concatenate([[0], np.cumsum(csz)[:-1]]).astype(np.uint64)
    dlen = np.full(len(csz), unit, np.uint64); dlen[-1] = n - unit * (len(csz) - 1)
    do = np.arange(len(csz), dtype=np.uint64) * unit
    perm = rng.permutation(len(csz))
    comp = torch.from_numpy(np.frombuffer(frames, dtype=np.uint8).copy()).cuda()
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    r, status, dsz = dctx.decompress_frames_device(out.data_ptr(), do[perm], dlen[perm], comp.data_ptr(), so[perm], csz[perm])
    assert r == n and not status.any() and (dsz == dlen[perm]).all()
    assert torch.equal(out.cpu(), torch.from_numpy(src))


def test_shim_decompress(env):
    z, lo, _, _ = env
    shim = C.CDLL(os.path.join(os.path.dirname(z.LIB_PATH), "libzstd_hipshim.so"))
    shim.ZSTD_compress.restype = C.c_size_t
    shim.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    shim.ZSTD_decompress.restype = C.c_size_t
    shim.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    shim.ZSTD_compressBound.restype = C.c_size_t
    shim.ZSTD_compressBound.argtypes = [C.c_size_t]
    shim.ZSTD_findDecompressedSize.restype = C.c_ulonglong
    shim.ZSTD_findDecompressedSize.argtypes = [C.c_void_p, C.c_size_t]
    shim.ZSTD_getFrameContentSize.restype = C.c_ulonglong
    shim.ZSTD_getFrameContentSize.argtypes = [C.c_void_p, C.c_size_t]
    shim.ZSTD_findFrameCompressedSize.restype = C.c_size_t
    shim.ZSTD_findFrameCompressedSize.argtypes = [C.c_void_p, C.c_size_t]
    shim.ZSTD_isError.restype = C.c_uint
    shim.ZSTD_isError.argtypes = [C.c_size_t]
    a = text_like(500000, 17)
    cap = shim.ZSTD_compressBound(len(a))
    dst = np.empty(cap, dtype=np.uint8)
    r = shim.ZSTD_compress(_buf(dst), cap, _buf(a), len(a), 3)
    assert not shim.ZSTD_isError(r)
    assert shim.ZSTD_findDecompressedSize(_buf(dst), r) == len(a)
    assert shim.ZSTD_getFrameContentSize(_buf(dst), r) == 131072
    first = shim.ZSTD_findFrameCompressedSize(_buf(dst), r)
    assert 0 < first < r
    out = np.empty(len(a), dtype=np.uint8)
    k = shim.ZSTD_decompress(_buf(out), len(a), _buf(dst), r)
    assert k == len(a) and (out == a).all()
    k = shim.ZSTD_decompress(_buf(out), len(a) - 1, _buf(dst), r)          # dstSize_tooSmall
    assert shim.ZSTD_isError(k) and (1 << 64) - k == 70


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (travels to the GPU box as a binary)")
def test_frames_without_content_size_small_windows_checksums(env):
    """zhip_decompress on frames whose header does not state the content size (decoded into bound-sized slots, then packed),
    small windows, checksums — alone and as a stream of several such frames"""
    z, lo, dctx, _ = env
    lr = load_ref()
    if not hasattr(lr, "zref_compress_frame_params"):
        pytest.skip("oracle/_ref predates zref_compress_frame_params")
    from test_oracle_decode import ref_frame_params
    cases = [(text_like(500000, 2), 3), (datagen(lo, 300000, 50, 3), 1), (text_like(40000, 4), 9), (np.zeros(200000, np.uint8), 3), (text_like(3, 5), 1)]
    stream, want = b"", b""
    for a, level in cases:
        for cs, ck, wl in ((0, 0, 0), (0, 1, 0), (1, 1, 10), (0, 1, 12), (1, 0, 16), (0, 0, 27)):
            f = ref_frame_params(lr, a, level, cs, ck, wl)
            assert dctx.decompress(f) == a.tobytes(), (len(a), level, cs, ck, wl)
            stream += f; want += a.tobytes()
    assert dctx.decompress(stream) == want


@pytest.mark.parametrize("checksum", [False, True])
def test_seekable_random_access(env, checksum):
    """zhip_seekable_read == ZSTD_seekable_decompress: random-access reads into a seekable file (frames + seek table) decode only
    the frames they touch; the same reads through the reference's seekable decoder where oracle/_ref travelled"""
    z, lo, dctx, _ = env
    ctx = z.Context(0, max_units=64)
    a = np.concatenate([datagen(lo, 131072 * 5 + 777, 50, 12), text_like(131072 * 4 + 4321, 3)])
    ctx.set_checksum(checksum)
    blob = ctx.compress_seekable(a, level=3)
    ctx.set_checksum(False)
    lr = load_ref() if have_ref() else None
    if lr is not None:
        lr.zref_seekable_read.restype = C.c_size_t
        lr.zref_seekable_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_ulonglong, C.c_void_p]
    src = np.frombuffer(blob, dtype=np.uint8)
    rng = np.random.default_rng(2)
    reads = [(0, 100), (131072 - 50, 100), (131072 * 3 + 17, 300000), (len(a) - 77, 77), (0, len(a)), (131072 * 2, 131072)] + [(int(o), 70000) for o in rng.integers(0, len(a) - 70000, size=8)]
    for off, ln in reads:
        got = dctx.seekable_read(blob, off, ln)
        assert got == a[off:off + ln].tobytes(), (off, ln)
        if lr is not None:
            out = np.zeros(ln, dtype=np.uint8)
            assert lr.zref_seekable_read(_buf(out), ln, _buf(src), len(src), off, None) == ln and out.tobytes() == got
    with pytest.raises(z.ZhipError):
        dctx.seekable_read(blob, len(a) - 10, 11)               # beyond the end
    with pytest.raises(z.ZhipError):
        dctx.seekable_read(blob[:-1], 0, 10)                    # no footer
    if checksum:
        bad = bytearray(blob); bad[len(blob) - 9 - 4] ^= 1      # the last frame's checksum in the table
        with pytest.raises(z.ZhipError, match="22"):
            dctx.seekable_read(bytes(bad), len(a) - 100, 100)


def test_oversized_raw_and_rle_blocks_like_the_one_shot_reference(env):
    """raw / RLE blocks above 128 KB: accepted, as ZSTD_decompress accepts them (tests/test_oracle_decode.py pins that to the reference)"""
    z = env[0] if isinstance(env, tuple) else env
    from test_oracle_decode import oversized_block_frames
    d = z.DContext()
    for frame, content in oversized_block_frames():
        assert d.decompress(frame) == content, len(content)
