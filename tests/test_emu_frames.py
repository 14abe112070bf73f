"""Multi-block frames on the host SIMT emulator: k_frame_fast against the oracle's
zo_compress_frame — one frame holding the blocks ZSTD_compress emits (lib/compress/zstd_compress.c:4520-4640)."""
import numpy as np
import pytest
from _libs import *
from _libs import UNIT_DT, _buf


@pytest.fixture(scope="module")
def libs():
    return load_oracle(), load_emu()


def _cases(lo):
    rng = np.random.default_rng(5)
    c = [("empty", np.zeros(0, np.uint8)), ("tiny", datagen(lo, 5, 50, 1)), ("small", datagen(lo, 5000, 50, 2)),
         ("one_block", datagen(lo, 131072, 50, 3)), ("block_plus_1", datagen(lo, 131073, 50, 4)),
         ("three_blocks", datagen(lo, 400000, 50, 5)), ("text", text_like(300000, 3)),
         ("random", rng.integers(0, 256, size=300000, dtype=np.uint8)),              # raw blocks: repcodes / tables not confirmed
         ("zeros", np.zeros(500000, np.uint8))]                                       # RLE blocks after the first
    m = np.concatenate([datagen(lo, 150000, 50, 1), rng.integers(0, 256, size=140000, dtype=np.uint8), np.full(200000, 7, np.uint8), text_like(150000, 9)])
    c.append(("mixed", m))
    return c


@pytest.mark.parametrize("level", [1, 3, -1, -5])
def test_frames_match_oracle(libs, level):
    lo, le = libs
    cases = _cases(lo)
    got = emu_compress_frames(le, lo, [a for _, a in cases], level)
    for (name, a), g in zip(cases, got):
        assert g == oracle_frame(lo, a, level), f"{name} level {level}"


def test_tiny_frames_match_oracle(libs):
    """frames of 1 .. 24 bytes in one batch (7 bytes is the size at which the parsers run with their search limit before the source:
    the oracle once wrapped there, tests/test_oracle_vs_reference.py::test_tiny_frames_vs_reference pins it to the reference)"""
    lo, le = libs
    rng = np.random.default_rng(8)
    bufs = [rng.integers(0, 256, size=n, dtype=np.uint8) for n in range(1, 25)] + [np.full(n, 66, np.uint8) for n in range(1, 25)]
    for level, cp in ((1, None), (3, None), (1, [17, 13, 17, 1, 7, 16, 1]), (1, [18, 12, 12, 1, 5, 0, 2])):
        got = emu_compress_frames(le, lo, bufs, level, cparams=cp)
        for a, g in zip(bufs, got):
            want = oracle_frame(lo, a, level) if cp is None else oracle_frame_params(lo, a, (C.c_uint * 7)(*cp), False)
            assert g == want, (len(a), level, cp)


def test_frames_table_in_hbm_and_checksum(libs):
    """level 2 above 256 KB: hashLog 16 -> the table is in HBM; the frame checksum is XXH64 of the WHOLE input"""
    lo, le = libs
    a = datagen(lo, 600000, 50, 11)
    got = emu_compress_frames(le, lo, [a], 2)[0]
    assert got == oracle_frame(lo, a, 2)
    ck = emu_compress_frames(le, lo, [a], 1, checksum=True)[0]
    plain = oracle_frame(lo, a, 1)
    assert ck[4] == plain[4] | 4 and ck[5:-4] == plain[5:]
    lo.zo_xxh64.restype = C.c_uint64
    lo.zo_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    assert int.from_bytes(ck[-4:], "little") == lo.zo_xxh64(a.ctypes.data_as(C.c_void_p), a.size, 0) & 0xFFFFFFFF


@pytest.mark.parametrize("level,js,ov,ck", [(1, 524288, 0, True), (3, 524288, 9, False)])
def test_job_pool_frames_match_oracle(libs, level, js, ov, ck):
    """ZSTD_c_nbWorkers semantics: k_frame_fast with a job table (prefix fill, zero repcodes, chunked block rule, one checksum)
    against zo_compress_frame_mt_params"""
    lo, le = libs
    rng = np.random.default_rng(8)
    cases = [("dg_1.1m", datagen(lo, 1_100_000, 50, 1)),
             ("mixed", np.concatenate([datagen(lo, 300000, 50, 3), rng.integers(0, 256, size=150000, dtype=np.uint8), np.zeros(300000, np.uint8), text_like(100000, 6)]))]
    for name, a in cases:
        assert emu_compress_frame_jobs(le, lo, a, level, js, ov, ck) == oracle_frame_mt(lo, a, level, js, ov, ck), (name, level, js, ov, ck)


def test_wave_checksum_kernel_matches_xxh64(libs):
    """k_xxh64_wave (one wavefront per large unit: staged pre-multiplied blocks + four accumulator chains) on sizes around its
    4 KB block and 32-byte stripe borders, at unaligned starts"""
    lo, le = libs
    lo.zo_xxh64.restype = C.c_uint64
    lo.zo_xxh64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
    le.emu_xxh64_wave.restype = None
    le.emu_xxh64_wave.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
    sizes = [0, 1, 31, 32, 33, 4095, 4096, 4097, 8191, 8192, 8192 + 37, 12288 + 4, 100_003, 1_000_000]
    rng = np.random.default_rng(12)
    src = rng.integers(0, 256, size=sum(sizes) + 3 * len(sizes) + 64, dtype=np.uint8)
    units = np.zeros(len(sizes), dtype=UNIT_DT)
    pos = 0
    for i, n in enumerate(sizes):
        pos += 3                                                         # odd offsets: the loads are unaligned
        units[i]["srcOff"] = pos; units[i]["srcLen"] = n
        pos += n
    chk = np.zeros(len(sizes) + 16, dtype=np.uint32)
    le.emu_xxh64_wave(_buf(src), _buf(units), len(sizes), _buf(chk), 0)
    for i, n in enumerate(sizes):
        o = int(units[i]["srcOff"])
        want = lo.zo_xxh64(src[o:].ctypes.data_as(C.c_void_p), n, 0) & 0xFFFFFFFF
        assert int(chk[i]) == want, n


@pytest.mark.parametrize("level,js,ov", [(2, 524288, 8), (3, 524288, 7)])
def test_job_window_starting_at_the_frames_first_byte(libs, level, js, ov):
    """jobSize <= overlap: the second job's prefix is everything before it, and the reference can match the frame's very first byte
    from it (its indices start at 2).  On the device a job counts from 1 with src one byte before its window; here that byte does
    not exist, and the parsers must not touch position 0 (tab_guard / non-empty candidates only).  HBM tables (hashLog >= 16)."""
    lo, le = libs
    a = datagen(lo, 700000, 20, 10).copy()
    a[js: js + 64] = a[0: 64]
    assert emu_compress_frame_jobs(le, lo, a, level, js, ov, False) == oracle_frame_mt(lo, a, level, js, ov, False)


def test_job_prefix_fill_counts_the_chain_log(libs):
    """ZSTD_loadDictionaryContent keeps the last 8 << max(hashLog, chainLog) bytes of a prefix (zstd_compress.c:4889-4896) — the chain
    log counts even for ZSTD_fast, which has no chain table (explicit parameters with chainLog > hashLog; found by the job fuzz)"""
    lo, le = libs
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.integers(0, 256, size=524288, dtype=np.uint8), np.zeros(30000, np.uint8)])
    a[524288:] = a[524288 - 200000: 524288 - 200000 + 30000]           # the second job repeats prefix bytes 200 000 back: inside 8 << 15, outside 8 << 14
    eff = (C.c_uint * 7)(19, 15, 14, 1, 4, 2, 1)
    want = oracle_frame_mt(lo, a, -2, 524288, 9, False, cp=eff)
    assert len(want) < 530000                                            # the match is there to be found
    assert emu_compress_frame_jobs(le, lo, a, -2, 524288, 9, False, cp=eff) == want
