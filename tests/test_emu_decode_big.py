"""The block-parallel decoder of ONE large frame (zstd_amd/csrc/zhip_decode_big.h: block walk, defining blocks of repeated tables,
symbolic repeat offsets + scan, copy map + pointer jumping) on the host SIMT emulator: frames of the oracle's compressor
(multi-block, every strategy: treeless literals, repeat-mode FSE tables, RLE / raw blocks) must come back byte for byte, and
damaged frames must either be declined (the library then falls back to k_decode) or decode to what the oracle's decoder makes
of the same bytes — never anything else, never a wild access."""
import numpy as np
import pytest
from _libs import *
from _libs import _buf


@pytest.fixture(scope="module")
def libs():
    lo, le = load_oracle(), load_emu()
    le.emu_decode_big.restype = C.c_uint
    le.emu_decode_big.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return lo, le


def big(le, frame, cap):
    src = np.frombuffer(bytes(frame) + b"\0" * 16, dtype=np.uint8).copy()
    dst = np.full(cap + 64, 0xEE, dtype=np.uint8)
    osz, ck, rd = C.c_uint(0), C.c_uint(0), C.c_uint(0)
    st = le.emu_decode_big(_buf(src), len(frame), _buf(dst), cap, C.byref(osz), C.byref(ck), C.byref(rd), 0)
    assert bytes(dst[cap:]) == b"\xEE" * 64                      # nothing behind the stated capacity was touched
    return st, dst[:osz.value].tobytes(), rd.value


def _inputs(lo):
    rng = np.random.default_rng(5)
    return [("dg_400k", datagen(lo, 400000, 50, 5)), ("text_300k", text_like(300000, 3)), ("zeros_500k", np.zeros(500000, np.uint8)),
            ("random_300k", rng.integers(0, 256, size=300000, dtype=np.uint8)),
            ("mixed", np.concatenate([datagen(lo, 150000, 50, 1), rng.integers(0, 256, size=140000, dtype=np.uint8), np.full(200000, 7, np.uint8), text_like(150000, 9)])),
            ("tiny", datagen(lo, 300, 50, 2)), ("period700", np.tile(rng.integers(0, 256, size=700, dtype=np.uint8), 600))]


def test_big_frames_round_trip(libs):
    lo, le = libs
    for name, a in _inputs(lo):
        for level in (1, 3):
            f = oracle_frame(lo, a, level)
            st, out, _ = big(le, f, len(a))
            assert st == 0 and out == a.tobytes(), (name, level)


def test_big_frames_of_the_lazy_strategies(libs):
    """greedy and above repeat FSE tables by cost: the decoder must rebuild them from the block that described them"""
    lo, le = libs
    a = np.concatenate([datagen(lo, 300000, 50, 5), text_like(200000, 8)])
    for level, row in ((5, 1), (7, 0), (9, 1)):
        cp = (C.c_uint * 7)()
        assert lo.zo_get_cparams(level, len(a), cp) == 0
        f = oracle_frame_params(lo, a, cp, row)
        modes = 0
        st, out, _ = big(le, f, len(a))
        assert st == 0 and out == a.tobytes(), (level, row)


def test_job_pool_frame_round_trip(libs):
    lo, le = libs
    a = datagen(lo, 1_300_000, 50, 1)
    f = oracle_frame_mt(lo, a, 1, 524288, 0, 0)
    st, out, _ = big(le, f, len(a))
    assert st == 0 and out == a.tobytes()


def test_declines_what_it_does_not_handle(libs):
    lo, le = libs
    a = datagen(lo, 200000, 50, 1)
    f = oracle_frame(lo, a, 1)
    assert big(le, f, len(a) - 1)[0] != 0                        # destination too small for the stated content
    assert big(le, f[:len(f) // 2], len(a))[0] != 0              # truncated
    assert big(le, b"\x00" * 64, 100)[0] != 0                     # not a frame


def test_damaged_frames_never_decode_to_something_else(libs):
    lo, le = libs
    rng = np.random.default_rng(11)
    a = np.concatenate([datagen(lo, 150000, 50, 7), text_like(120000, 4)])
    base = bytearray(oracle_frame(lo, a, 3))
    declined = same = 0
    for trial in range(60):
        f = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(6, len(f)))
            f[p] ^= 1 << int(rng.integers(0, 8))
        st, out, _ = big(le, bytes(f), len(a))
        want = oracle_decompress(lo, bytes(f), len(a))
        if st != 0:
            declined += 1
        else:
            assert want is not None and out == want, trial
            same += 1
    assert declined > 0


@pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the reference built by oracle/Makefile)")
def test_frames_of_the_reference_at_every_strategy(libs):
    """the REAL reference's frames — levels -3 .. 22: every strategy up to btultra2, its block splitter's variable blocks, its table modes —
    through the block-parallel decoder"""
    lo, le = libs
    lr = load_ref()
    lr.zref_compress_frame.restype = C.c_size_t
    lr.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    rng = np.random.default_rng(2)
    a = np.concatenate([datagen(lo, 200000, 70, 1), rng.integers(0, 256, size=100000, dtype=np.uint8), np.zeros(150000, np.uint8), text_like(200000, 9),
                        np.tile(rng.integers(0, 256, size=333, dtype=np.uint8), 500)])
    for level in (-3, 1, 3, 6, 9, 13, 16, 19, 22):
        d = np.zeros(len(a) + (len(a) >> 7) + 1024, dtype=np.uint8)
        k = lr.zref_compress_frame(level, _buf(a), len(a), _buf(d), len(d))
        assert k != ERR
        st, out, _ = big(le, d[:k].tobytes(), len(a))
        assert st == 0 and out == a.tobytes(), level


def _strip_content_size(frame):
    """the same frame as a streaming compressor without a pledged size writes it: no content size in the header, a window descriptor instead"""
    f = bytes(frame)
    fhd = f[4]
    single, fcs_code = (fhd >> 5) & 1, fhd >> 6
    fcs_b = (1 if single else 0) if fcs_code == 0 else (1 << fcs_code)
    pos = 5 + (0 if single else 1)
    v = int.from_bytes(f[pos: pos + fcs_b], "little") + (256 if fcs_code == 1 else 0)
    wd = f[5] if not single else (max(10, int(v - 1).bit_length() if v > 1 else 10) - 10) << 3
    return f[:4] + bytes([fhd & 0x04, wd]) + f[pos + fcs_b:]


def test_frames_that_do_not_state_their_content_size(libs):
    lo, le = libs
    rng = np.random.default_rng(9)
    a = np.concatenate([datagen(lo, 300000, 50, 2), text_like(150000, 3), rng.integers(0, 256, size=50000, dtype=np.uint8)])
    for level in (1, 3):
        f = _strip_content_size(oracle_frame(lo, a, level))
        assert oracle_decompress(lo, f, len(a) + 1000) == a.tobytes()          # still a valid frame
        st, out, _ = big(le, f, len(a) + 1000)                                  # the destination slot is a bound, not the size
        assert st == 0 and out == a.tobytes(), level
        assert big(le, f, len(a) - 1)[0] != 0                                   # too small a slot is declined
