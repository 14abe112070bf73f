"""The device decoder (zstd_amd/csrc/zhip_decode.h, unmodified) on the host SIMT emulator, checked against the oracle decoder
and the committed decode vectors: the wave-level logic (bit readers, table builds, producer/consumer hand-over, lane-parallel
copies) without a GPU."""
import base64, ctypes as C, hashlib, json, os, zlib
import numpy as np
import pytest
from _libs import load_oracle, load_emu, load_ref, have_ref, _buf, ERR, datagen, text_like, oracle_decompress, corpus_cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_v1.json")
DFRAME_DT = np.dtype([("srcOff", "<u8"), ("dstOff", "<u8"), ("srcLen", "<u4"), ("dstCap", "<u4")])
DRESULT_DT = np.dtype([("status", "<u4"), ("size", "<u4"), ("hasChecksum", "<u4"), ("checksum", "<u4")])


def unpack(s):
    return zlib.decompress(base64.b64decode(s))


@pytest.fixture(scope="module")
def libs():
    lo, le = load_oracle(), load_emu()
    assert le.emu_sizeof_dframe() == DFRAME_DT.itemsize
    le.emu_decode.restype = C.c_int
    le.emu_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_int]
    return lo, le


def emu_decode(le, frames, caps, dictionary=None, groups=0):
    """frames: list of bytes, caps: room per frame -> list of (status, bytes)"""
    src = np.frombuffer(b"".join(frames) + b"\x00" * 16, dtype=np.uint8).copy()
    fr = np.zeros(len(frames), dtype=DFRAME_DT)
    so = do = 0
    for i, (f, c) in enumerate(zip(frames, caps)):
        fr[i] = (so, do, len(f), c)
        so += len(f); do += c
    dst = np.full(do + 64, 0xEE, dtype=np.uint8)
    res = np.zeros(len(frames), dtype=DRESULT_DT)
    d = None if dictionary is None else np.frombuffer(bytes(dictionary), dtype=np.uint8)
    e = le.emu_decode(_buf(src), _buf(fr), len(frames), _buf(dst), None if d is None else _buf(d), 0 if d is None else len(d),
                      _buf(res), groups, 0)
    assert e == 0
    out = []
    lo = load_oracle()
    lo.zo_xxh64.restype = C.c_uint64
    lo.zo_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    for i in range(len(frames)):
        o = int(fr["dstOff"][i])
        st, data = int(res["status"][i]), dst[o:o + int(res["size"][i])]
        # the product verifies content checksums with two more kernels (k_xxh64 over the output + k_dec_verify); the same check here
        if st == 0 and int(res["hasChecksum"][i]):
            x = lo.zo_xxh64(data.ctypes.data_as(C.c_void_p) if len(data) else None, len(data), 0) & 0xFFFFFFFF
            if x != int(res["checksum"][i]):
                st, data = 0xC5, data[:0]
        out.append((st, data.tobytes(), res[i]))
    assert (dst[do:] == 0xEE).all(), "wrote past the destination"
    return out


def test_golden_vectors_on_the_emulator(libs):
    lo, le = libs
    g = json.load(open(GOLD))
    vs = g["fixtures"] + g["frames"]
    frames = [unpack(v["zst"]) for v in vs]
    got = emu_decode(le, frames, [v["size"] for v in vs], groups=3)
    for v, (st, data, _) in zip(vs, got):
        assert st == 0, (v["name"], st)
        assert len(data) == v["size"] and hashlib.sha256(data).hexdigest() == v["sha256"], v["name"]
    bad = emu_decode(le, [unpack(v["zst"]) for v in g["errors"]], [4096] * len(g["errors"]))
    for v, (st, _, _) in zip(g["errors"], bad):
        assert st != 0, v["name"]


def test_own_frames_round_trip(libs):
    """frames of the oracle's compressor (= the product's byte stream) at all implemented levels, ragged sizes"""
    lo, _ = libs
    le = libs[1]
    frames, want = [], []
    for n in (0, 1, 6, 7, 64, 1000, 4097, 70000, 131072):
        for name, a in (corpus_cases(lo, sizes=(n,), seeds=(2,)) if n else [("empty", np.zeros(0, dtype=np.uint8))]):
            if n >= 70000 and not any(k in name for k in ("P50", "text", "zeros", "random", "period7", "farmatch", "halfrun", "lowent")):
                continue
            for level in (1, 3, 5):
                cap = lo.zo_compress_bound(max(n, 1)) + 64
                dst = np.empty(cap, dtype=np.uint8)
                r = lo.zo_compress_unit(_buf(dst), cap, _buf(a) if n else None, n, level)
                assert r != ERR
                frames.append(dst[:r].tobytes()); want.append(a.tobytes())
    got = emu_decode(le, frames, [len(w) for w in want], groups=0)
    for i, ((st, data, _), w) in enumerate(zip(got, want)):
        assert st == 0 and data == w, (i, st, len(w))


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_reference_frames_and_corruptions(libs):
    lo, le = libs
    lr = load_ref()
    from test_oracle_decode import ref_frame
    rng = np.random.default_rng(21)
    frames, want = [], []
    for kind, n, level in (("text", 300000, 3), ("P50", 200000, 1), ("text", 50000, 19), ("P80", 140000, -3), ("text", 9000, 7), ("P50", 131072, 12)):
        a = text_like(n, 5) if kind == "text" else datagen(lo, n, int(kind[1:]), 6)
        frames.append(ref_frame(lr, a, level)); want.append(a.tobytes())
    got = emu_decode(le, frames, [len(w) for w in want])
    for (st, data, _), w in zip(got, want):
        assert st == 0 and data == w
    # corrupted frames: the device rejects whatever the oracle rejects and decodes alike otherwise
    bases = [ref_frame(lr, text_like(5000, 9), 3), ref_frame(lr, datagen(lo, 7000, 50, 3), 1), ref_frame(lr, text_like(2500, 4), 19),
             ref_frame(lr, (rng.geometric(0.25, size=6000) % 256).astype(np.uint8), 5), ref_frame(lr, text_like(300, 6), 1)]
    muts = []
    for it in range(int(os.environ.get("ZHIP_EMU_MUTANTS", "300"))):
        b = bytearray(bases[it % len(bases)])
        k = rng.integers(0, 3)
        if k == 0:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        else:
            del b[int(rng.integers(9, len(b))):]
        muts.append(bytes(b))
    got = emu_decode(le, muts, [8192] * len(muts))
    nerr = 0
    for m, (st, data, _) in zip(muts, got):
        w = oracle_decompress(lo, m, 8192)
        if w is None:
            assert st != 0, m.hex()[:60]
            nerr += 1
        else:
            assert st == 0 and data == w, (st, m.hex()[:60])
    assert nerr > 30


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_dictionary_frames_on_the_emulator(libs):
    lo, le = libs
    lr = load_ref()
    zd = np.fromfile(os.path.join(os.path.dirname(GOLD), "github_like_110k.zdict"), dtype=np.uint8)
    rng = np.random.default_rng(8)
    raw = text_like(20000, 31)
    for dict_ in (zd, raw):
        recs = []
        for i in range(24):
            n = int(rng.integers(20, 2500))
            st = int(rng.integers(0, len(dict_) - n))
            r = dict_[st:st + n].copy()
            r[rng.integers(0, n, size=max(1, n // 40))] = rng.integers(32, 127, size=max(1, n // 40), dtype=np.uint8)
            recs.append(r)
        src = np.concatenate(recs)
        sizes = (C.c_size_t * len(recs))(*[len(r) for r in recs])
        outs = (C.c_size_t * len(recs))()
        cap = int(lr.zref_compress_bound(len(src))) + 64 * len(recs)
        dst = np.empty(cap, dtype=np.uint8)
        tot = lr.zref_compress_records_cdict(3, _buf(dict_), len(dict_), _buf(src), sizes, len(recs), _buf(dst), cap, outs)
        assert tot != ERR
        frames, off = [], 0
        for i in range(len(recs)):
            frames.append(dst[off:off + outs[i]].tobytes()); off += outs[i]
        got = emu_decode(le, frames, [len(r) for r in recs], dictionary=dict_.tobytes(), groups=2)
        for (st, data, _), r in zip(got, recs):
            assert st == 0 and data == r.tobytes()
        if dict_ is zd:                                        # the frames name their dictionary: decoding without it is refused
            st, _, _ = emu_decode(le, frames[:1], [4096])[0]
            assert st == 32


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_frame_parameter_variants_on_the_emulator(libs):
    """no content size in the header, small windows (hundreds of small blocks per frame), checksums"""
    lo, le = libs
    lr = load_ref()
    if not hasattr(lr, "zref_compress_frame_params"):
        pytest.skip("oracle/_ref predates zref_compress_frame_params")
    from test_oracle_decode import ref_frame_params
    frames, want = [], []
    for a, level in ((text_like(150000, 2), 3), (datagen(lo, 90000, 50, 3), 1), (np.zeros(70000, np.uint8), 3), (text_like(3, 5), 1)):
        for cs, ck, wl in ((0, 0, 0), (0, 1, 0), (1, 1, 10), (0, 1, 12), (1, 0, 16)):
            frames.append(ref_frame_params(lr, a, level, cs, ck, wl)); want.append(a.tobytes())
    got = emu_decode(le, frames, [len(w) + 5 for w in want], groups=4)
    for (st, data, res), w in zip(got, want):
        assert st == 0 and data == w


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_corrupted_multi_block_frames(libs):
    """mutants of multi-block frames (repeat modes, a frame without content size + checksum, a small window, RLE blocks): the device
    rejects whatever the oracle rejects — the oracle is pinned to the reference on exactly this (tests/test_oracle_decode.py) — and
    decodes alike otherwise; nothing is written past the destination"""
    lo, le = libs
    lr = load_ref()
    from test_oracle_decode import ref_frame, ref_frame_params
    rng = np.random.default_rng(5)
    bases = [ref_frame(lr, text_like(300000, 2), 3), ref_frame(lr, datagen(lo, 400000, 50, 3), 1), ref_frame(lr, text_like(200000, 4), 19),
             ref_frame_params(lr, datagen(lo, 300000, 70, 5), 5, 0, 1, 17), ref_frame_params(lr, text_like(150000, 6), 9, 0, 0, 12),
             ref_frame(lr, np.zeros(300000, np.uint8), 3)]
    cap = 420000
    muts = []
    for it in range(int(os.environ.get("ZHIP_EMU_MUTANTS_BIG", "96"))):
        b = bytearray(bases[it % len(bases)])
        k = rng.integers(0, 4)
        if k == 0:
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        elif k == 1:
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif k == 2:
            del b[int(rng.integers(9, len(b))):]
        else:
            b[int(rng.integers(0, min(len(b), 64)))] = int(rng.integers(0, 256))          # the headers
        muts.append(bytes(b))
    nerr = nok = 0
    for i in range(0, len(muts), 24):
        chunk = muts[i:i + 24]
        for m, (st, data, _) in zip(chunk, emu_decode(le, chunk, [cap] * len(chunk))):
            w = oracle_decompress(lo, m, cap)
            if w is None:
                assert st != 0, m[:40].hex()
                nerr += 1
            else:
                assert st == 0 and data == w, (st, m[:40].hex())
                nok += 1
    assert nerr > 20 and nok > 5
