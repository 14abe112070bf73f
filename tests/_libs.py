"""ctypes loaders for the TEST-ONLY checkers: oracle/libzoracle.so (our C restatement) and
oracle/_ref/libzref_shim.so (the real reference, prebuilt from /root/reference by oracle/Makefile)."""
import ctypes as C
import os
import sys
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ERR = C.c_size_t(-1).value
u8p = C.POINTER(C.c_uint8)


def _buf(a):
    return a.ctypes.data_as(C.c_void_p)


def load_oracle():
    so = os.path.join(ORACLE_DIR, "libzoracle.so")
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(os.path.join(ORACLE_DIR, f))
                                     for f in ("zoracle.c", "zoracle_dec.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libzoracle.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.zo_compress_bound.restype = C.c_size_t
    lib.zo_compress_bound.argtypes = [C.c_size_t]
    lib.zo_compress_unit.restype = C.c_size_t
    lib.zo_compress_unit.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    lib.zo_compress_unit_params.restype = C.c_size_t
    lib.zo_compress_unit_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.zo_compress_chunks.restype = C.c_size_t
    lib.zo_compress_chunks.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                       C.c_void_p, C.c_size_t]
    lib.zo_get_cparams.restype = C.c_int
    lib.zo_get_cparams.argtypes = [C.c_int, C.c_ulonglong, C.c_void_p]
    lib.zo_sequences_public.restype = C.c_size_t
    lib.zo_sequences_public.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zo_parse_block.restype = C.c_size_t
    lib.zo_parse_block.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
    lib.zo_datagen.restype = None
    lib.zo_datagen.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_uint]
    lib.zo_hist.restype = C.c_size_t
    lib.zo_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.zo_huf_build.restype = C.c_uint
    lib.zo_huf_build.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    lib.zo_fse_normalize.restype = C.c_int
    lib.zo_fse_normalize.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint]
    lib.zo_compress_literals.restype = C.c_size_t
    lib.zo_compress_literals.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
    # decoder restatement (oracle/zoracle_dec.c)
    lib.zo_decompress.restype = C.c_size_t
    lib.zo_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zo_decompress_dict.restype = C.c_size_t
    lib.zo_decompress_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zo_frame_info.restype = C.c_int
    lib.zo_frame_info.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    return lib


def oracle_decompress(lo, frames, cap, dictionary=None):
    """bytes -> bytes through the oracle decoder, or None when it reports an error"""
    src = np.frombuffer(bytes(frames), dtype=np.uint8)
    dst = np.empty(max(cap, 1), dtype=np.uint8)
    if dictionary is None:
        r = lo.zo_decompress(_buf(dst), cap, _buf(src), len(src))
    else:
        d = np.frombuffer(bytes(dictionary), dtype=np.uint8)
        r = lo.zo_decompress_dict(_buf(dst), cap, _buf(src), len(src), _buf(d), len(d))
    return None if r == ERR else dst[:r].tobytes()


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libzref_shim.so"))


def load_ref():
    lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libzref_shim.so"))
    lib.zref_compress_chunks.restype = C.c_size_t
    lib.zref_compress_chunks.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                         C.c_void_p, C.c_size_t]
    lib.zref_compress_chunks_norow.restype = C.c_size_t
    lib.zref_compress_chunks_norow.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_size_t]
    lib.zref_compress_chunks_params.restype = C.c_size_t
    lib.zref_compress_chunks_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                                C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zref_compress_frame.restype = C.c_size_t
    lib.zref_compress_frame.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zref_sequences.restype = C.c_size_t
    lib.zref_sequences.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zref_decompress.restype = C.c_size_t
    lib.zref_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zref_decompress_dict.restype = C.c_size_t
    lib.zref_decompress_dict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zref_compress_records_cdict.restype = C.c_size_t
    lib.zref_compress_records_cdict.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                                C.c_void_p, C.c_size_t, C.c_void_p]
    if hasattr(lib, "zref_compress_frame_params"):
        lib.zref_compress_frame_params.restype = C.c_size_t
        lib.zref_compress_frame_params.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.zref_decompressed_size.restype = C.c_ulonglong
    lib.zref_decompressed_size.argtypes = [C.c_void_p, C.c_size_t]
    lib.zref_compress_bound.restype = C.c_size_t
    lib.zref_compress_bound.argtypes = [C.c_size_t]
    lib.zref_get_cparams.restype = None
    lib.zref_get_cparams.argtypes = [C.c_int, C.c_ulonglong, C.c_size_t, C.c_void_p]
    lib.zref_datagen.restype = None
    lib.zref_datagen.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_uint]
    lib.zref_lorem.restype = None
    lib.zref_lorem.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    lib.zref_hist.restype = C.c_size_t
    lib.zref_hist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.zref_huf_build.restype = C.c_size_t
    lib.zref_huf_build.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    lib.zref_huf_compress.restype = C.c_size_t
    lib.zref_huf_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    lib.zref_fse_normalize.restype = C.c_size_t
    lib.zref_fse_normalize.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint]
    lib.zref_fse_optimal_tablelog.restype = C.c_uint
    lib.zref_fse_optimal_tablelog.argtypes = [C.c_uint, C.c_size_t, C.c_uint]
    return lib


# ----------------------------------------------------------------------------- corpus (seeded, no files needed)
def datagen(lib_o, n, P, seed=0, lit=0.0):
    a = np.zeros(max(n, 1), dtype=np.uint8)
    lib_o.zo_datagen(_buf(a), n, P / 100.0, lit, seed)
    return a[:n]


def lorem(lr, n, seed=0):
    """what `zstd -b#` compresses when it is given no file: LOREM_genBuffer(buffer, n, seed) (programs/lorem.h:20, programs/benchzstd.c:1014), made by the
    reference's own generator as oracle/_ref/libzstd_ref.so holds it (lr = load_ref(): the shim resolves the symbol through its dependency)"""
    lr.LOREM_genBuffer.restype = None
    lr.LOREM_genBuffer.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    a = np.zeros(max(n, 1), dtype=np.uint8)
    lr.LOREM_genBuffer(_buf(a), n, seed)
    return a[:n]


def text_like(n, seed):
    """word-salad text (a lorem-ish stand-in that needs no reference code)."""
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(300)]
    out = bytearray()
    p = rng.zipf(1.3, size=n // 3 + 16) % len(words)
    i = 0
    while len(out) < n:
        out += words[p[i]] + (b". " if i % 11 == 10 else b" ")
        i += 1
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


def corpus_cases(lib_o, sizes=(131072,), seeds=(0,)):
    """yield (name, np.uint8 array) covering compressible / incompressible / degenerate inputs."""
    for n in sizes:
        for s in seeds:
            for P in (0, 10, 20, 50, 80, 95, 100):
                yield f"datagen_P{P}_n{n}_s{s}", datagen(lib_o, n, P, s)
            yield f"text_n{n}_s{s}", text_like(n, s)
            rng = np.random.default_rng(1000 + s)
            yield f"random_n{n}_s{s}", rng.integers(0, 256, size=n, dtype=np.uint8)
            yield f"lowent_n{n}_s{s}", rng.integers(0, 4, size=n, dtype=np.uint8)
            yield f"skew_n{n}_s{s}", (rng.geometric(0.3, size=n) % 256).astype(np.uint8)
            yield f"zeros_n{n}", np.zeros(n, dtype=np.uint8)
            yield f"period7_n{n}", (np.arange(n) % 7).astype(np.uint8)
            yield f"ramp_n{n}", (np.arange(n) // 3 % 256).astype(np.uint8)
            if n >= 4096:
                a = rng.integers(0, 256, size=n, dtype=np.uint8)
                a[n // 3: n // 3 + n // 4] = a[: n // 4]        # one long far match
                yield f"farmatch_n{n}_s{s}", a
                b = datagen(lib_o, n, 50, s + 7).copy()
                b[n // 2:] = 65                                  # long literal-free tail run
                yield f"halfrun_n{n}_s{s}", b


# ----------------------------------------------------------------------------- host SIMT emulator of the product kernels
UNIT_DT = np.dtype([("srcOff", "<u8"), ("srcLen", "<u4"), ("windowLog", "u1"), ("chainLog", "u1"), ("hashLog", "u1"),
                    ("minMatch", "u1"), ("strategy", "u1"), ("searchLog", "u1"), ("litMode", "u1"), ("pad0", "u1"),
                    ("targetLength", "<u4"), ("rowLog", "<u4"), ("pad1", "<u4")])
SEQ_DT = np.dtype([("offBase", "<u4"), ("litLength", "<u2"), ("mlBase", "<u2")])
PARSE_DT = np.dtype([("nbSeq", "<u4"), ("lastLits", "<u4"), ("longPos", "<u4"), ("longType", "<u4"),
                     ("rep", "<u4", (3,)), ("status", "<u4"), ("litSize", "<u4"), ("pad0", "<u4")])


def load_emu():
    d = os.path.join(ROOT, "tests", "simt")
    so = os.path.join(d, "libzhip_emu.so")
    srcs = [os.path.join(d, "emu_driver.cpp"), os.path.join(d, "simt_runtime.cpp"), os.path.join(d, "hip", "hip_runtime.h")]
    srcs += [os.path.join(ROOT, "zstd_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "zstd_amd", "csrc"))
             if f.endswith(".h")]
    extra = os.environ.get("ZHIP_EMU_FLAGS", "").split()          # debugging: another build of the emulator library (e.g. -DZHIP_WIN_FAST=0)
    if extra:
        so = os.path.join(d, "libzhip_emu_dbg.so")
    if extra or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-I" + d,
                               "-I" + os.path.join(ROOT, "zstd_amd", "csrc"), srcs[0], srcs[1], "-o", so, "-lpthread"] + extra)
    lib = C.CDLL(so)
    assert lib.emu_sizeof_unit() == UNIT_DT.itemsize and lib.emu_sizeof_parse() == PARSE_DT.itemsize
    lib.emu_parse_fast.restype = None
    lib.emu_parse_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int]
    lib.emu_lit_stride.restype = C.c_uint
    return lib


def make_units(lo, sizes, level, unit=131072, row=False):
    """unit table for buffers laid out back to back; sizes = list of unit lengths.  row: greedy / lazy / lazy2 units with
    windowLog > 14 use the row-hash matcher (the reference's default), else the hash chain"""
    units = np.zeros(len(sizes), dtype=UNIT_DT)
    off = 0
    for i, n in enumerate(sizes):
        cp = (C.c_uint * 7)()
        assert lo.zo_get_cparams(level, n, cp) == 0
        rowLog = min(6, max(4, cp[3])) if (row and 3 <= cp[6] <= 5 and cp[0] > 14) else 0
        units[i] = (off, n, cp[0], cp[1], cp[2], cp[4], cp[6], cp[3], 1 if (cp[6] == 1 and cp[5] > 0) else 0, 0, cp[5], rowLog, 0)
        off += n
    return units


def oracle_parse(lo, a, level, want_lits=False):
    """-> (seqs[nb,3] = litLength, matchLength, offBase ; litSize, rep[3] [, literal bytes])"""
    n = len(a)
    cp = (C.c_uint * 7)()
    assert lo.zo_get_cparams(level, n, cp) == 0
    cap = n // 3 + 8
    seqs = np.zeros((cap, 3), dtype=np.uint32)
    lits = np.zeros(n + 64, dtype=np.uint8)
    litSize = C.c_size_t(0)
    rep = (C.c_uint * 3)()
    nb = lo.zo_parse_block(cp, _buf(a), n, _buf(seqs), cap, _buf(lits), C.byref(litSize), rep)
    assert nb != ERR
    if want_lits:
        return seqs[:nb].copy(), litSize.value, list(rep), lits[:litSize.value].copy()
    return seqs[:nb].copy(), litSize.value, list(rep)


def emu_parse_units(le, src, units, seqs, lits, metas):
    """stage 1 on the emulator: fast or dfast kernel according to the units' strategy (all units alike)"""
    nu = len(units)
    strat = int(units["strategy"][0]) if nu else 1
    assert (units["strategy"] == strat).all() or ((units["strategy"] >= 3) & (units["strategy"] <= 5)).all()
    if strat == 1:
        smem = le.emu_fast_lds_bytes(int(units["hashLog"].max()))
        qmode = int(os.environ.get("ZHIP_EMU_QUEUE", "0"))
        if qmode and nu:
            # the queue form (k_order_cost + k_order_sort + k_parse_fast_q / k_parse_fast_g): LDS tables, global tables, or both on one queue
            le.emu_parse_fast_queue.restype = None
            le.emu_parse_fast_queue.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_int, C.c_void_p, C.c_int]
            order = np.zeros(nu + 1, dtype=np.uint32)
            le.emu_parse_fast_queue(_buf(src), _buf(units), nu, _buf(seqs), _buf(lits), _buf(metas), smem, qmode, _buf(order), 0)
            assert sorted(order[:nu].tolist()) == list(range(nu)), "the dispatch order is not a permutation"
        else:
            le.emu_parse_fast(_buf(src), _buf(units), nu, _buf(seqs), _buf(lits), _buf(metas), smem, 0)
    elif strat >= 3:
        le.emu_hc_table_words.restype = C.c_uint64
        stride = (int(le.emu_hc_table_words(int(units["hashLog"].max()))) + 3) & ~3
        tabs = np.full(nu * stride + 4, 0xEEEEEEEE, dtype=np.uint32)
        best = np.full(nu * 131072 + 4, 0xEEEEEEEEEEEEEEEE, dtype=np.uint64)
        le.emu_parse_lazy.restype = None
        le.emu_parse_lazy.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        le.emu_parse_lazy(_buf(src), _buf(units), nu, _buf(tabs), stride, _buf(best), _buf(seqs), _buf(lits), _buf(metas), 0)
    else:
        le.emu_dfast_table_bytes.restype = C.c_uint64
        stride = max(int(le.emu_dfast_table_bytes(int(h), int(c))) for h, c in zip(units["hashLog"], units["chainLog"])) // 4
        stride = (stride + 3) & ~3
        tabs = np.full(nu * stride + 4, 0xEEEEEEEE, dtype=np.uint32)
        le.emu_parse_dfast.restype = None
        le.emu_parse_dfast.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        le.emu_parse_dfast(_buf(src), _buf(units), nu, _buf(tabs), stride, _buf(seqs), _buf(lits), _buf(metas), 0)


def emu_compress_units(le, lo, bufs, level, checksum=False, row=False):
    """run stage 1 + stage 2 of the product kernels on the emulator; returns list of frame bytes"""
    le.emu_entropy.restype = None
    le.emu_entropy.argtypes = [C.c_void_p] * 2 + [C.c_uint] + [C.c_void_p] * 6 + [C.c_int]
    sizes = [len(b) for b in bufs]
    units = make_units(lo, sizes, level, row=row)
    src = np.concatenate(list(bufs) + [np.zeros(16, dtype=np.uint8)])
    cap = le.emu_seq_cap()
    nu = len(bufs)
    seqs = np.zeros(nu * cap, dtype=SEQ_DT)
    metas = np.zeros(nu, dtype=PARSE_DT)
    ostride, lstride = le.emu_out_stride(), le.emu_lit_stride()
    lits = np.full(nu * lstride, 0xEE, dtype=np.uint8)
    emu_parse_units(le, src, units, seqs, lits, metas)
    stb = np.full(nu * 3 * cap, 0xEEEE, dtype=np.uint16)
    out = np.full(nu * ostride, 0xEE, dtype=np.uint8)
    osz = np.zeros(nu, dtype=np.uint32)
    if checksum:
        chk = np.zeros(nu + 16, dtype=np.uint32)
        le.emu_xxh64.restype = None
        le.emu_xxh64.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        le.emu_xxh64(_buf(src), _buf(units), nu, _buf(chk), 0)
        le.emu_entropy_ck.restype = None
        le.emu_entropy_ck.argtypes = [C.c_void_p] * 2 + [C.c_uint] + [C.c_void_p] * 7 + [C.c_int]
        le.emu_entropy_ck(_buf(src), _buf(units), nu, _buf(seqs), _buf(metas), _buf(lits), _buf(stb), _buf(out), _buf(osz), _buf(chk), 0)
    else:
        le.emu_entropy(_buf(src), _buf(units), nu, _buf(seqs), _buf(metas), _buf(lits), _buf(stb), _buf(out), _buf(osz), 0)
    return [out[i * ostride: i * ostride + int(osz[i])].tobytes() for i in range(nu)]


def oracle_frame(lo, a, level):
    """the oracle's multi-block single frame (strategy fast) for one input"""
    lo.zo_frame_bound.restype = C.c_size_t; lo.zo_frame_bound.argtypes = [C.c_size_t]
    lo.zo_compress_frame.restype = C.c_size_t
    lo.zo_compress_frame.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    cap = lo.zo_frame_bound(len(a))
    dst = np.zeros(cap, dtype=np.uint8)
    r = lo.zo_compress_frame(_buf(dst), cap, _buf(a) if len(a) else None, len(a), level)
    assert r != ERR
    return dst[:r].tobytes()


def emu_compress_frames(le, lo, bufs, level, checksum=False, cparams=None):
    """the frame kernel (one workgroup per multi-block frame) on the emulator; returns list of frame bytes
    (cparams: effective parameters [windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy] for every frame)"""
    sizes = [len(b) for b in bufs]
    frames = make_units(lo, sizes, level)                  # parameters come from the whole input's size class
    if cparams is not None:
        cp = list(cparams)
        for f in frames:
            f["windowLog"], f["chainLog"], f["hashLog"], f["searchLog"], f["minMatch"], f["targetLength"], f["strategy"] = cp
            f["litMode"] = 1 if (cp[6] == 1 and cp[5] > 0) else 0
    src = np.concatenate(list(bufs) + [np.zeros(16, dtype=np.uint8)])
    nf = len(bufs)
    ostride = (max(sizes) + (max(sizes) >> 8) + 2048 + 15) & ~15
    out = np.full(nf * ostride, 0xEE, dtype=np.uint8)
    osz = np.zeros(nf, dtype=np.uint32)
    chk = None
    if checksum:
        chk = np.zeros(nf + 16, dtype=np.uint32)
        le.emu_xxh64_wave.restype = None
        le.emu_xxh64_wave.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        le.emu_xxh64_wave(_buf(src), _buf(frames), nf, _buf(chk), 0)
    le.emu_frame_fast.restype = None
    le.emu_frame_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
    le.emu_frame_fast(_buf(src), _buf(frames), nf, _buf(out), ostride, _buf(osz), _buf(chk) if checksum else None, 0)
    return [out[i * ostride: i * ostride + int(osz[i])].tobytes() for i in range(nf)]


def frame_cases(lo):
    """inputs for the multi-block frame tests (tests/golden/frames_v1.json pins the reference's output for each)"""
    rng = np.random.default_rng(77)
    yield "dg_131073", datagen(lo, 131073, 50, 4)
    yield "dg_400000", datagen(lo, 400000, 50, 5)
    yield "dg_1m", datagen(lo, 1 << 20, 50, 6)
    yield "dg_p90_3m", datagen(lo, 3 << 20, 90, 7)                    # > 2^19: matches past the level-1 window are refused
    yield "text_700k", text_like(700000, 3)
    yield "random_300k", rng.integers(0, 256, size=300000, dtype=np.uint8)
    yield "zeros_500k", np.zeros(500000, np.uint8)
    yield "lowent_400k", rng.integers(0, 4, size=400000, dtype=np.uint8)
    yield "mixed_640k", np.concatenate([datagen(lo, 150000, 50, 1), rng.integers(0, 256, size=140000, dtype=np.uint8),
                                        np.full(200000, 7, np.uint8), text_like(150000, 9)])
    b = datagen(lo, 262144 + 50, 50, 8).copy(); b[131072:262144] = b[:131072]      # a block that is one long match into the previous one
    yield "repeat_block", b
    yield "tail_6", datagen(lo, 131072 + 6, 50, 9)                    # last block below the 7-byte minimum


JOB_DT = np.dtype([("start", "<u4"), ("prefixLen", "<u4"), ("flags", "<u4"), ("ownHeader", "<u4"), ("frameSize", "<u8"), ("frameIdx", "<u4"), ("pad0", "<u4")])


def frame_header_bytes(n, window_log):
    single = (1 << window_log) >= n
    fcs = (n >= 256) + (n >= 65536 + 256)
    return 4 + 1 + (0 if single else 1) + ((1 if single else 0) if fcs == 0 else (2 if fcs == 1 else 4))


def make_jobs(lo, n, level, job_size=0, overlap_log=0, cp=None):
    """the job table of one frame compressed with ZSTD_c_nbWorkers >= 1 (zstdmt_compress.c: sections of the job size, each later one
    with the overlap as prefix) -> (units, jobs); n must exceed 512 KB (below that the reference does not use jobs)"""
    if cp is None:
        cp = (C.c_uint * 7)()
        assert lo.zo_get_cparams(level, n, cp) == 0
    lo.zo_mt_job_size.restype = C.c_size_t; lo.zo_mt_job_size.argtypes = [C.c_void_p, C.c_ulonglong]
    lo.zo_mt_overlap_size.restype = C.c_size_t; lo.zo_mt_overlap_size.argtypes = [C.c_void_p, C.c_int]
    sec, ov = lo.zo_mt_job_size(cp, job_size), lo.zo_mt_overlap_size(cp, overlap_log)
    sec = max(sec, ov)
    starts = list(range(0, n, sec))
    units = np.zeros(len(starts), dtype=UNIT_DT); jobs = np.zeros(len(starts), dtype=JOB_DT)
    prev = 0
    for k, s in enumerate(starts):
        ln = min(sec, n - s)
        units[k] = (0, ln, cp[0], cp[1], cp[2], cp[4], cp[6], cp[3], 1 if (cp[6] == 1 and cp[5] > 0) else 0, 0, cp[5], 0, 0)
        pre = 0 if k == 0 else min(prev, ov)
        jobs[k] = (s, pre, (1 if k == 0 else 0) | (2 if s + ln == n else 0), frame_header_bytes(ln, cp[0]), n, 0, 0)
        prev = ln
    return units, jobs, cp


def emu_compress_frame_jobs(le, lo, a, level, job_size=0, overlap_log=0, checksum=False, cp=None):
    """one frame as parallel jobs on the emulator; returns the frame bytes (cp: effective parameters instead of the level's)"""
    n = len(a)
    units, jobs, cp = make_jobs(lo, n, level, job_size, overlap_log, cp)
    assert le.emu_sizeof_job() == JOB_DT.itemsize
    src = np.concatenate([a, np.zeros(16, dtype=np.uint8)])
    nj = len(units)
    ostride = (int(units["srcLen"].max()) + (int(units["srcLen"].max()) >> 8) + 2048 + 15) & ~15
    out = np.full(nj * ostride, 0xEE, dtype=np.uint8); osz = np.zeros(nj, dtype=np.uint32)
    chk = None
    if checksum:
        whole = np.zeros(1, dtype=UNIT_DT); whole[0] = units[0]; whole[0]["srcLen"] = n
        chk = np.zeros(17, dtype=np.uint32)
        le.emu_xxh64_wave.restype = None
        le.emu_xxh64_wave.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        le.emu_xxh64_wave(_buf(src), _buf(whole), 1, _buf(chk), 0)
    le.emu_frame_jobs.restype = None
    le.emu_frame_jobs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
    le.emu_frame_jobs(_buf(src), _buf(units), _buf(jobs), nj, _buf(out), ostride, _buf(osz), _buf(chk) if checksum else None, 0)
    return b"".join(out[i * ostride: i * ostride + int(osz[i])].tobytes() for i in range(nj))


def oracle_frame_mt(lo, a, level, job_size=0, overlap_log=0, checksum=False, cp=None):
    lo.zo_compress_frame_mt_params.restype = C.c_size_t
    lo.zo_compress_frame_mt_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int]
    lo.zo_frame_bound.restype = C.c_size_t; lo.zo_frame_bound.argtypes = [C.c_size_t]
    if cp is None:
        cp = (C.c_uint * 7)()
        assert lo.zo_get_cparams(level, len(a), cp) == 0
    cap = lo.zo_frame_bound(len(a)) + 4
    dst = np.zeros(cap, dtype=np.uint8)
    r = lo.zo_compress_frame_mt_params(_buf(dst), cap, _buf(a), len(a), cp, job_size, overlap_log, 1 if checksum else 0)
    assert r != ERR
    return dst[:r].tobytes()


# (level, jobSize, overlapLog, checksum) of the ZSTD_c_nbWorkers tests; jobSize 0 / overlapLog 0 = the reference's defaults
MT_MODES = [(1, 0, 0, 0), (1, 524288, 0, 1), (1, 700000, 9, 0), (1, 524288, 1, 0), (3, 0, 0, 0), (3, 1 << 20, 3, 1), (-1, 524288, 0, 0), (2, 600000, 8, 0)]


def mt_frame_cases(lo):
    """inputs above 512 KB for the job-pool frames (tests/golden/frames_mt_v1.json pins the reference's output for each)"""
    rng = np.random.default_rng(99)
    yield "dg_524289", datagen(lo, 524289, 50, 14)                    # one byte above ZSTDMT_JOBSIZE_MIN: two jobs, the second 1 byte
    yield "dg_5m", datagen(lo, 5 << 20, 50, 15)
    yield "dg_p80_2.3m", datagen(lo, 2_300_001, 80, 16)
    yield "text_3m", text_like(3_000_000, 13)
    yield "random_1.2m", rng.integers(0, 256, size=1_200_000, dtype=np.uint8)
    yield "zeros_2m", np.zeros(2 << 20, np.uint8)                     # RLE blocks: the rule differs for a job's first block
    yield "mixed_2.6m", np.concatenate([datagen(lo, 700000, 50, 1), rng.integers(0, 256, size=500000, dtype=np.uint8),
                                        np.full(600000, 7, np.uint8), text_like(800000, 9)])
    b = datagen(lo, 3 << 19, 50, 18).copy(); b[1 << 19: 2 << 19] = b[: 1 << 19]; b[2 << 19:] = b[: 1 << 19]    # sections that repeat earlier ones: the overlap matters
    yield "repeat_sections", b


def ref_frame_mt(lr, a, level, job_size=0, overlap_log=0, checksum=False, cp=None):
    lr.zref_compress_frame_mt.restype = C.c_size_t
    lr.zref_compress_frame_mt.argtypes = [C.c_int, C.c_void_p, C.c_ulonglong, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    dst = np.zeros(len(a) + (len(a) >> 7) + 1024, dtype=np.uint8)
    cpi = (C.c_int * 7)(*cp) if cp is not None else None
    r = lr.zref_compress_frame_mt(level, cpi, job_size, overlap_log, 1 if checksum else 0, _buf(a), len(a), _buf(dst), len(dst))
    assert r != ERR
    return dst[:r].tobytes()


def lazy_frame_cases(lo):
    """inputs for the multi-block frames of the lazy strategies (tests/golden/frames_lazy_v1.json pins the reference's output)"""
    rng = np.random.default_rng(55)
    yield "dg_400000", datagen(lo, 400000, 50, 5)
    yield "dg_p85_2.5m", datagen(lo, 2_500_000, 85, 7)               # beyond the chain table, long matches: the 384 / 192 nextToUpdate rule at block starts
    yield "text_700k", text_like(700000, 3)
    yield "random_300k", rng.integers(0, 256, size=300000, dtype=np.uint8)
    yield "zeros_500k", np.zeros(500000, np.uint8)
    yield "mixed_900k", np.concatenate([datagen(lo, 250000, 50, 1), rng.integers(0, 256, size=200000, dtype=np.uint8),
                                        np.full(200000, 7, np.uint8), text_like(250000, 9)])     # the fingerprint splitter of lazy2 sees borders
    yield "tail_10", datagen(lo, 131072 + 10, 50, 9)                 # a last block below the row matcher's 16-byte guard


# (level, noRow) of the lazy-frame golden: greedy 5, lazy 6-7, lazy2 8-10; the row matcher (the reference's default) and the hash chain
LAZY_FRAME_MODES = [(5, 0), (5, 1), (6, 0), (7, 1), (8, 0), (9, 1), (10, 0)]


def oracle_frame_params(lo, a, cp, row):
    lo.zo_compress_frame_params.restype = C.c_size_t
    lo.zo_compress_frame_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_frame_bound.restype = C.c_size_t; lo.zo_frame_bound.argtypes = [C.c_size_t]
    lo.zo_set_row_matcher(1 if row else 0)
    try:
        cap = lo.zo_frame_bound(len(a))
        dst = np.zeros(cap, dtype=np.uint8)
        r = lo.zo_compress_frame_params(_buf(dst), cap, _buf(a), len(a), cp)
        assert r != ERR
        return dst[:r].tobytes()
    finally:
        lo.zo_set_row_matcher(0)


def lazy_dict_cases(lo):
    """(name, dictionary bytes, records) for the lazy-strategy CDict oracle (tests/golden/dict_lazy_v1.json)"""
    sys.path.insert(0, ROOT)
    from zstd_amd import workloads as W
    zd = np.fromfile(os.path.join(ROOT, "tests", "golden", "github_like_110k.zdict"), dtype=np.uint8)
    flat, offs = W.github_like_records(40, seed=9)
    yield "zdict_json", zd, [flat[int(offs[i]):int(offs[i + 1])].copy() for i in range(len(offs) - 1)]
    rng = np.random.default_rng(31)
    corpus = np.concatenate([text_like(150000, 3), datagen(lo, 150000, 50, 4)])
    recs = []
    for n in (0, 1, 7, 8, 15, 16, 17, 40, 300, 1200, 5000, 20000, 32768, 32769, 50000, 131072):     # above 32 KB: the copy mode
        o = int(rng.integers(20000, len(corpus) - n))
        r = corpus[o:o + n].copy()
        if n > 20:
            r[n // 2: n // 2 + 5] = rng.integers(0, 256, 5)
        recs.append(r)
    yield "raw_20k", corpus[:20000].copy(), recs


def oracle_records_cdict(lo, d, recs, level, row):
    """every record as its own frame with a CDict of `level` (attach mode), concatenated"""
    lo.zo_cdict_create.restype = C.c_void_p; lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
    lo.zo_cdict_free.argtypes = [C.c_void_p]
    lo.zo_compress_unit_cdict.restype = C.c_size_t; lo.zo_compress_unit_cdict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_compress_bound.restype = C.c_size_t; lo.zo_compress_bound.argtypes = [C.c_size_t]
    lo.zo_set_row_matcher(1 if row else 0)
    try:
        cd = lo.zo_cdict_create(_buf(d), len(d), level)
        assert cd
        out = []
        for r in recs:
            o = np.zeros(lo.zo_compress_bound(len(r)) + 64, dtype=np.uint8)
            k = lo.zo_compress_unit_cdict(_buf(o), len(o), _buf(r) if len(r) else None, len(r), cd)
            assert k != ERR
            out.append(o[:k].tobytes())
        lo.zo_cdict_free(cd)
        return out
    finally:
        lo.zo_set_row_matcher(0)


def row_log_of(cp, row):
    """ZSTD_resolveRowMatchFinderMode + rowLog (zstd_compress.c:237-253, :2042): 0 = the hash chain"""
    if not row or cp[0] <= 14 or not 3 <= cp[6] <= 5:
        return 0
    return min(max(cp[3], 4), 6)


def emu_compress_frames_lazy(le, lo, bufs, cps, row, checksum=False, threads=0):
    """k_lz_links + k_lz_search + k_frame_lazy on the emulator: one multi-block frame per input with the lazy strategies;
    cps[i] = effective parameters of frame i [windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy]"""
    sizes = [len(b) for b in bufs]
    nf = len(bufs)
    units = np.zeros(nf, dtype=UNIT_DT)
    off = 0
    for i, (n, cp) in enumerate(zip(sizes, cps)):
        units[i] = (off, n, cp[0], cp[1], cp[2], cp[4], cp[6], cp[3], 0, 0, cp[5], row_log_of(cp, row), 0)
        off += n
    src = np.concatenate(list(bufs) + [np.zeros(16, dtype=np.uint8)])
    ostride = (max(sizes) + (max(sizes) >> 8) + 2048 + 15) & ~15
    out = np.full(nf * ostride, 0xEE, dtype=np.uint8); osz = np.zeros(nf, dtype=np.uint32)
    chk = None
    if checksum:
        chk = np.zeros(nf + 16, dtype=np.uint32)
        le.emu_xxh64_wave.restype = None
        le.emu_xxh64_wave.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        le.emu_xxh64_wave(_buf(src), _buf(units), nf, _buf(chk), 0)
    le.emu_frame_lazy.restype = None
    le.emu_frame_lazy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
    le.emu_frame_lazy(_buf(src), _buf(units), None, nf, _buf(out), ostride, _buf(osz), _buf(chk) if checksum else None, threads)
    return [out[i * ostride: i * ostride + int(osz[i])].tobytes() for i in range(nf)]


def emu_compress_frame_jobs_lazy(le, lo, a, cp, row, job_size=0, overlap_log=0, checksum=False, threads=0):
    """one frame of a lazy strategy as parallel jobs (ZSTD_c_nbWorkers semantics) on the emulator; returns the frame bytes"""
    n = len(a)
    units, jobs, cp = make_jobs(lo, n, 0, job_size, overlap_log, cp)
    units["rowLog"] = row_log_of(cp, row)
    src = np.concatenate([a, np.zeros(16, dtype=np.uint8)])
    nj = len(units)
    ostride = (int(units["srcLen"].max()) + (int(units["srcLen"].max()) >> 8) + 2048 + 15) & ~15
    out = np.full(nj * ostride, 0xEE, dtype=np.uint8); osz = np.zeros(nj, dtype=np.uint32)
    chk = None
    if checksum:
        whole = np.zeros(1, dtype=UNIT_DT); whole[0] = units[0]; whole[0]["srcLen"] = n
        chk = np.zeros(17, dtype=np.uint32)
        le.emu_xxh64_wave.restype = None
        le.emu_xxh64_wave.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_int]
        le.emu_xxh64_wave(_buf(src), _buf(whole), 1, _buf(chk), 0)
    le.emu_frame_lazy.restype = None
    le.emu_frame_lazy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int]
    le.emu_frame_lazy(_buf(src), _buf(units), _buf(jobs), nj, _buf(out), ostride, _buf(osz), _buf(chk) if checksum else None, threads)
    return b"".join(out[i * ostride: i * ostride + int(osz[i])].tobytes() for i in range(nj))
