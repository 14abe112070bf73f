"""The decoder restatement (oracle/zoracle_dec.c) pinned to the real reference decoder and to the committed decode vectors.

* with oracle/_ref present (this container): frames made by the real reference at many levels / sizes (single- and
  multi-block, checksummed, dictionary) decode to the input and to what ZSTD_decompress returns; random corruptions the
  reference rejects are rejected (and decode alike when both accept);
* everywhere: tests/golden/decode_v1.json — the reference's own golden-decompression fixtures, its
  golden-decompression-errors, and reference-made frames with the sha256 of their content.
"""
import base64, ctypes as C, hashlib, json, os, zlib
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, _buf, ERR, datagen, text_like, oracle_decompress, corpus_cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode_v1.json")


def unpack(s):
    return zlib.decompress(base64.b64decode(s))


@pytest.fixture(scope="module")
def lo():
    return load_oracle()


def test_golden_decode_vectors(lo):
    g = json.load(open(GOLD))
    assert len(g["fixtures"]) == 4 and len(g["errors"]) == 3 and len(g["frames"]) >= 12
    for v in g["fixtures"] + g["frames"]:
        out = oracle_decompress(lo, unpack(v["zst"]), v["size"] + 64)
        assert out is not None, v["name"]
        assert len(out) == v["size"] and hashlib.sha256(out).hexdigest() == v["sha256"], v["name"]
    for v in g["errors"]:
        assert oracle_decompress(lo, unpack(v["zst"]), 1 << 20) is None, v["name"]


def test_frame_info(lo):
    g = json.load(open(GOLD))
    for v in g["frames"]:
        z = np.frombuffer(unpack(v["zst"]) + b"\x00" * 7, dtype=np.uint8)
        cs, ds = C.c_size_t(0), C.c_ulonglong(0)
        assert lo.zo_frame_info(_buf(z), len(z), C.byref(cs), C.byref(ds)) == 0
        assert cs.value == len(z) - 7 and ds.value == v["size"], v["name"]


def ref_frame(lr, a, level):
    cap = int(lr.zref_compress_bound(len(a)))
    dst = np.empty(cap, dtype=np.uint8)
    r = lr.zref_compress_frame(level, _buf(a), len(a), _buf(dst), cap)
    assert r != ERR
    return dst[:r].tobytes()


def ref_decompress(lr, z, cap):
    src = np.frombuffer(bytes(z), dtype=np.uint8)
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    r = lr.zref_decompress(_buf(dst), cap, _buf(src), len(src))
    return None if r == ERR else dst[:r].tobytes()


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_oracle_decodes_reference_frames(lo):
    lr = load_ref()
    n_cases = 0
    for n in (0, 1, 7, 100, 1000, 20000, 131072, 131073, 400000):
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(1,)) if n else [("empty", np.zeros(0, dtype=np.uint8))]:
            for level in (1, 3, 6, -3) + ((19,) if n <= 20000 else ()) + ((12,) if n in (131072, 400000) and "text" in name else ()):
                z = ref_frame(lr, a, level)
                out = oracle_decompress(lo, z, n + 32)
                assert out == a.tobytes(), (name, n, level)
                n_cases += 1
    # concatenated frames + a skippable frame in between
    a, b = text_like(50000, 3), datagen(lo, 70000, 50, 4)
    z = ref_frame(lr, a, 3) + b"\x50\x2a\x4d\x18\x05\x00\x00\x00hello" + ref_frame(lr, b, 1)
    assert oracle_decompress(lo, z, 200000) == a.tobytes() + b.tobytes()
    assert oracle_decompress(lo, z + b"\x00", 200000) is None          # trailing garbage: srcSize_wrong
    assert n_cases > 300


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_checksummed_frames(lo):
    lr = load_ref()
    lr.zref_compress_chunks_checksum.restype = C.c_size_t
    lr.zref_compress_chunks_checksum.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    a = text_like(300000, 9)
    cap = int(lr.zref_compress_bound(len(a))) + 4096
    dst = np.empty(cap, dtype=np.uint8)
    r = lr.zref_compress_chunks_checksum(3, 131072, _buf(a), len(a), _buf(dst), cap, None, 0)
    assert r != ERR
    z = dst[:r].tobytes()
    assert oracle_decompress(lo, z, len(a)) == a.tobytes()
    bad = bytearray(z); bad[-1] ^= 1
    assert oracle_decompress(lo, bad, len(a)) is None and ref_decompress(lr, bad, len(a)) is None


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_dictionary_frames(lo):
    lr = load_ref()
    zd = np.fromfile(os.path.join(os.path.dirname(GOLD), "github_like_110k.zdict"), dtype=np.uint8)
    rng = np.random.default_rng(5)
    raw = text_like(30000, 77)
    for dict_ in (zd, raw):
        recs = []
        for i in range(40):
            n = int(rng.integers(20, 3000))
            st = int(rng.integers(0, len(dict_) - n))
            r = dict_[st:st + n].copy()
            r[rng.integers(0, n, size=max(1, n // 40))] = rng.integers(32, 127, size=max(1, n // 40), dtype=np.uint8)
            recs.append(r)
        src = np.concatenate(recs)
        sizes = (C.c_size_t * len(recs))(*[len(r) for r in recs])
        outs = (C.c_size_t * len(recs))()
        cap = int(lr.zref_compress_bound(len(src))) + 64 * len(recs)
        dst = np.empty(cap, dtype=np.uint8)
        for level in (1, 3):
            tot = lr.zref_compress_records_cdict(level, _buf(dict_), len(dict_), _buf(src), sizes, len(recs), _buf(dst), cap, outs)
            assert tot != ERR
            off = 0
            for i, r in enumerate(recs):
                f = dst[off:off + outs[i]].tobytes(); off += outs[i]
                assert oracle_decompress(lo, f, len(r) + 8, dictionary=dict_.tobytes()) == r.tobytes(), (i, level)
            # without the dictionary the ZDICT frames carry a dictID -> rejected, like the reference does
            if dict_ is zd:
                assert oracle_decompress(lo, dst[:outs[0]].tobytes(), 4096) is None


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_corrupted_frames_agree_with_reference(lo):
    lr = load_ref()
    rng = np.random.default_rng(11)
    bases = [ref_frame(lr, text_like(6000, 1), 3), ref_frame(lr, datagen(lo, 9000, 50, 2), 1), ref_frame(lr, text_like(3000, 5), 19),
             ref_frame(lr, (rng.geometric(0.3, size=5000) % 256).astype(np.uint8), 5), ref_frame(lr, text_like(140000, 8), 1)[:0] or ref_frame(lr, text_like(1500, 8), 7)]
    agree_err = agree_ok = lenient = 0
    for z in bases:
        for _ in range(400):
            b = bytearray(z)
            kind = rng.integers(0, 3)
            if kind == 0:
                b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            else:
                del b[int(rng.integers(4, len(b))):]
            if len(b) == 0:
                continue
            want = ref_decompress(lr, b, 1 << 18)
            got = oracle_decompress(lo, b, 1 << 18)
            # The reference's 4-stream Huffman fast loop (huf_decompress.c:700-900, HUF_decompress4X1_usingDTable_internal_fast)
            # only checks the produced length, not that each bitstream was consumed exactly, so it accepts some corrupted
            # literal streams that its own strict loop (:600-700, BIT_endOfDStream) — which the oracle restates — rejects.
            # Required: the oracle never accepts what the reference rejects, and agrees on the bytes when both accept.
            if want is None:
                assert got is None, (bytes(b).hex()[:80], kind)
                agree_err += 1
            elif got is None:
                lenient += 1
            else:
                assert want == got
                agree_ok += 1
    assert agree_err > 500 and agree_ok > 20 and lenient < agree_err // 3, (agree_err, agree_ok, lenient)


def ref_frame_params(lr, a, level, content_size=1, checksum=0, window_log=0):
    cap = int(lr.zref_compress_bound(len(a))) + 64
    dst = np.empty(cap, dtype=np.uint8)
    r = lr.zref_compress_frame_params(level, content_size, checksum, window_log, _buf(a), len(a), _buf(dst), cap)
    assert r != ERR
    return dst[:r].tobytes()


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_frames_without_content_size_small_windows_checksums(lo):
    """frame-parameter variants of the real reference: no content size in the header (window descriptor instead), windows smaller
    than the content (blocks of windowSize bytes, offsets up to the window), checksum on — all decode like the reference"""
    lr = load_ref()
    if not hasattr(lr, "zref_compress_frame_params"):
        pytest.skip("oracle/_ref predates zref_compress_frame_params")
    cases = [(text_like(500000, 2), 3), (datagen(lo, 300000, 50, 3), 1), (text_like(40000, 4), 9), (np.zeros(200000, np.uint8), 3), (text_like(3, 5), 1)]
    for a, level in cases:
        for cs, ck, wl in ((0, 0, 0), (0, 1, 0), (1, 1, 10), (0, 1, 12), (1, 0, 16), (0, 0, 27)):
            z = ref_frame_params(lr, a, level, cs, ck, wl)
            assert oracle_decompress(lo, z, len(a) + 8) == a.tobytes(), (len(a), level, cs, ck, wl)
            cs_, ds_ = C.c_size_t(0), C.c_ulonglong(0)
            zb = np.frombuffer(z, dtype=np.uint8)
            assert lo.zo_frame_info(_buf(zb), len(zb), C.byref(cs_), C.byref(ds_)) == 0 and cs_.value == len(z)
            assert ds_.value == (len(a) if cs else 2**64 - 1)


def oversized_block_frames():
    """hand-built frames whose raw / RLE blocks exceed the 128 KB block maximum: ZSTD_decompress (one-shot, zstd_decompress.c:1012-1024)
    only checks them against the destination, unlike ZSTD_decompressContinue (:1313, :1365) — the one-shot behaviour is the contract"""
    rng = np.random.default_rng(77)
    out = []
    for n, kind in ((200000, "raw"), (200000, "rle"), (131073, "raw"), (2000000, "rle")):
        hdr = bytes([0x28, 0xB5, 0x2F, 0xFD, (2 << 6) | (1 << 5)]) + n.to_bytes(4, "little")          # single segment, 4-byte content size
        if kind == "raw":
            payload = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
            out.append((hdr + (1 | (0 << 1) | (n << 3)).to_bytes(3, "little") + payload, payload))
        else:
            out.append((hdr + (1 | (1 << 1) | (n << 3)).to_bytes(3, "little") + b"\x07", b"\x07" * n))
    return out


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the real reference)")
def test_oversized_raw_and_rle_blocks_like_the_one_shot_reference(lo):
    lr = load_ref()
    for frame, content in oversized_block_frames():
        want = ref_decompress(lr, bytearray(frame), len(content) + 16)
        got = oracle_decompress(lo, bytearray(frame), len(content) + 16)
        assert want == content and got == content, len(content)
