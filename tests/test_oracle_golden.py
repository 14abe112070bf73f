"""Oracle vs committed golden vectors (made by tests/golden/make_golden.py with the real reference).
Runs anywhere (no /root/reference, no oracle/_ref needed)."""
import ctypes as C, hashlib, json, os
import numpy as np
from _libs import datagen, load_oracle, corpus_cases, frame_cases, oracle_frame, mt_frame_cases, oracle_frame_mt, MT_MODES, _buf, ERR

GOLD = os.path.join(os.path.dirname(__file__), "golden", "units_v1.json")


GOLD_HC = os.path.join(os.path.dirname(__file__), "golden", "units_v2_hashchain.json")


def test_oracle_reproduces_golden_units():
    _check(GOLD, (1, 3), 400)


def test_oracle_reproduces_golden_hashchain_units():
    """greedy / lazy / lazy2 with the hash-chain matcher (the reference run with useRowMatchFinder disabled)"""
    _check(GOLD_HC, (5, 6, 7), 600)


GOLD_ROW = os.path.join(os.path.dirname(__file__), "golden", "units_v3_rowhash.json")


def test_oracle_reproduces_golden_rowhash_units():
    """greedy / lazy / lazy2 with the reference's DEFAULT matcher (row hash when windowLog > 14, fresh-CCtx salt)"""
    import ctypes as C
    lo = load_oracle()
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    lo.zo_set_row_matcher(1)
    try:
        gold = {(g["case"], g["level"]): g for g in json.load(open(GOLD_ROW))["units"]}
        seen = 0
        for n in (131072, 100001, 40000, 20000):
            for name, a in corpus_cases(lo, sizes=(n,), seeds=(0,)):
                for level in (5, 6, 7, 8, 9, 10):
                    g = gold.get((name, level))
                    if g is None:
                        continue
                    assert hashlib.sha256(a.tobytes()).hexdigest() == g["src_sha256"], name
                    dst = np.zeros(n + 1024, dtype=np.uint8)
                    r = lo.zo_compress_unit(_buf(dst), len(dst), _buf(a), n, level)
                    assert r != ERR and r == g["csize"] and hashlib.sha256(dst[:r].tobytes()).hexdigest() == g["dst_sha256"], (name, level)
                    seen += 1
        assert seen == len(gold) > 300
    finally:
        lo.zo_set_row_matcher(0)


def _check(path, levels, atleast):
    lo = load_oracle()
    gold = {(g["case"], g["level"]): g for g in json.load(open(path))["units"]}
    sizes = sorted({g["n"] for g in gold.values()})
    seen = 0
    for n in sizes:
        for name, a in corpus_cases(lo, sizes=(n,), seeds=(0, 5)):
            for level in levels:
                g = gold.get((name, level))
                if g is None:
                    continue
                assert hashlib.sha256(a.tobytes()).hexdigest() == g["src_sha256"], name
                cap = lo.zo_compress_bound(n) + 64
                dst = np.zeros(cap, dtype=np.uint8)
                r = lo.zo_compress_unit(_buf(dst), cap, _buf(a), n, level)
                assert r != ERR and r == g["csize"], (name, level)
                assert hashlib.sha256(dst[:r].tobytes()).hexdigest() == g["dst_sha256"], (name, level)
                seen += 1
    assert seen >= len(gold) > atleast


def test_oracle_reproduces_golden_dictionary_frames():
    """dictionary path (CDict attach mode, raw-content + ZDICT-trained): the oracle vs fixtures made with the real reference"""
    import ctypes as C, sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from zstd_amd import workloads as W
    from _libs import text_like
    lo = load_oracle()
    lo.zo_cdict_create.restype = C.c_void_p
    lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lo.zo_cdict_free.argtypes = [C.c_void_p]
    lo.zo_compress_unit_cdict.restype = C.c_size_t
    lo.zo_compress_unit_cdict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dict_v1.json")))["cases"]
    zd = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "github_like_110k.zdict"), dtype=np.uint8)
    raw = W.github_like_records(100, seed=5)[0][:60000].copy()
    flat, offs = W.github_like_records(120, seed=31)
    recs = [flat[int(offs[i]):int(offs[i + 1])].copy() for i in range(120)]
    t = text_like(40000, 31)
    recs += [np.zeros(0, np.uint8), recs[0][:6], recs[1][:7], recs[2][:8], recs[3][:9], recs[4][:100], np.concatenate(recs[5:12])[:8000], t[:3000], t[3000:3300]]
    assert len(gold) == 6
    for g in gold:
        d = zd if g["dict"] == "zdict" else raw
        assert hashlib.sha256(d.tobytes()).hexdigest() == g["dict_sha256"]
        assert hashlib.sha256(np.concatenate(recs).tobytes()).hexdigest() == g["records_sha256"]
        cd = lo.zo_cdict_create(_buf(d), len(d), g["level"])
        assert cd
        frames = b""
        for r, want in zip(recs, g["frame_sizes"]):
            buf = np.zeros(len(r) + 700, dtype=np.uint8)
            k = lo.zo_compress_unit_cdict(_buf(buf), len(buf), _buf(r), len(r), cd)
            assert k == want, (g["dict"], g["level"], len(r))
            frames += buf[:k].tobytes()
        assert hashlib.sha256(frames).hexdigest() == g["frames_sha256"], (g["dict"], g["level"])
        lo.zo_cdict_free(cd)


GOLD_FRAMES = os.path.join(os.path.dirname(__file__), "golden", "frames_v1.json")


def test_oracle_reproduces_golden_multiblock_frames():
    """inputs above 128 KB as ONE frame (zo_compress_frame; zstd_compress.c:4520-4640): block split at 128 KB / 92 KB, shared
    table and window, repcodes and Huffman table confirmed per compressed block, RLE and raw blocks"""
    lo = load_oracle()
    gold = {(g["case"], g["level"]): g for g in json.load(open(GOLD_FRAMES))["frames"]}
    seen = 0
    for name, a in frame_cases(lo):
        for level in (1, 2, 3, 4, -1, -5):
            g = gold.get((name, level))
            if g is None:
                continue
            assert hashlib.sha256(a.tobytes()).hexdigest() == g["src_sha256"], name
            out = oracle_frame(lo, a, level)
            assert len(out) == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], (name, level)
            seen += 1
    assert seen == len(gold) >= 60


GOLD_FRAMES_MT = os.path.join(os.path.dirname(__file__), "golden", "frames_mt_v1.json")


def test_oracle_reproduces_golden_job_pool_frames():
    """ZSTD_c_nbWorkers = 1 (zo_compress_frame_mt_params; zstdmt_compress.c): jobs of the job size, each a fresh context that loaded
    the overlap as prefix (every third position, repcodes zero), 512 KB chunks per compressContinue, `savings` counting the job's
    own discarded frame header, one checksum of the whole input"""
    lo = load_oracle()
    gold = {(g["case"], g["level"], g["jobSize"], g["overlapLog"], g["checksum"]): g for g in json.load(open(GOLD_FRAMES_MT))["frames"]}
    seen = 0
    for name, a in mt_frame_cases(lo):
        for level, js, ov, ck in MT_MODES:
            g = gold[(name, level, js, ov, ck)]
            assert hashlib.sha256(a.tobytes()).hexdigest() == g["src_sha256"], name
            out = oracle_frame_mt(lo, a, level, js, ov, bool(ck))
            assert len(out) == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], (name, level, js, ov, ck)
            seen += 1
    assert seen == len(gold) == 64


GOLD_FRAMES_LAZY = os.path.join(os.path.dirname(__file__), "golden", "frames_lazy_v1.json")


def test_oracle_reproduces_golden_lazy_multiblock_frames():
    """greedy / lazy / lazy2 across the blocks of one frame (zo_lazy_block: hash chain or rows, nextToUpdate and the window carried from
    block to block, the 384 / 192 catch-up rule at block starts; FSE tables repeated by cost, zstd_compress_sequences.c:205-231; the
    fingerprint block splitter of lazy2, zstd_preSplit.c) — the oracle side of the next frame kernel"""
    from _libs import lazy_frame_cases, LAZY_FRAME_MODES, oracle_frame_params
    import ctypes as C
    lo = load_oracle()
    gold = {(g["case"], g["level"], g["noRow"]): g for g in json.load(open(GOLD_FRAMES_LAZY))["frames"]}
    seen = 0
    for name, a in lazy_frame_cases(lo):
        for level, no_row in LAZY_FRAME_MODES:
            g = gold[(name, level, no_row)]
            assert hashlib.sha256(a.tobytes()).hexdigest() == g["src_sha256"], name
            cp = (C.c_uint * 7)()
            assert lo.zo_get_cparams(level, len(a), cp) == 0
            out = oracle_frame_params(lo, a, cp, row=not no_row)
            assert len(out) == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], (name, level, no_row)
            seen += 1
    assert seen == len(gold) == 49


def test_oracle_reproduces_golden_lazy_cdict_records():
    """records with a CDict whose parameter row is greedy / lazy / lazy2 (attach mode: ZSTD_compressBlock_*_dictMatchState, the working
    context's own hash chain or rows first, then the dictionary's — zstd_lazy.c:742-770, :1296-1334 — two-segment repcodes and catch-up;
    the dictionary's FSE tables repeated by cost): the oracle side of DESIGN.md §9 item 6"""
    from _libs import lazy_dict_cases, oracle_records_cdict
    lo = load_oracle()
    gold = {(g["case"], g["level"], g["noRow"]): g for g in json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dict_lazy_v1.json")))["streams"]}
    seen = 0
    for name, d, recs in lazy_dict_cases(lo):
        for level in (5, 6, 8, 10):
            for no_row in (0, 1):
                g = gold[(name, level, no_row)]
                out = b"".join(oracle_records_cdict(lo, d, recs, level, row=not no_row))
                assert len(out) == g["csize"] and hashlib.sha256(out).hexdigest() == g["dst_sha256"], (name, level, no_row)
                seen += 1
    assert seen == len(gold) == 16


def test_oracle_refuses_frames_whose_window_is_below_the_block_size():
    """with windowLog < 17 the reference's blocks follow the window (zstd_compress.c:4494-4501 blockSizeMax) and the window slides inside
    the parser: neither the oracle nor the device restates that — both refuse (zhip_lib.hip: "windowLog below the block size"), found by
    the emulator fuzz of the job-pool frames straying out of the domain"""
    lo = load_oracle()
    lo.zo_compress_frame_params.restype = C.c_size_t
    lo.zo_compress_frame_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_compress_frame_mt_params.restype = C.c_size_t
    lo.zo_compress_frame_mt_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int]
    lo.zo_frame_bound.restype = C.c_size_t; lo.zo_frame_bound.argtypes = [C.c_size_t]
    a = datagen(lo, 700000, 50, 3)
    cap = lo.zo_frame_bound(len(a)) + 4
    dst = np.zeros(cap, dtype=np.uint8)
    for wl in (11, 16):
        cp = (C.c_uint * 7)(wl, 10, 10, 1, 4, 0, 1)
        assert lo.zo_compress_frame_params(_buf(dst), cap, _buf(a), len(a), cp) == ERR
        assert lo.zo_compress_frame_mt_params(_buf(dst), cap, _buf(a), len(a), cp, 0, 0, 0) == ERR
        assert lo.zo_compress_frame_params(_buf(dst), cap, _buf(a), 1 << wl, cp) != ERR       # a source that fits the window is one block at most
    cp = (C.c_uint * 7)(17, 10, 10, 1, 4, 0, 1)
    assert lo.zo_compress_frame_params(_buf(dst), cap, _buf(a), len(a), cp) != ERR
