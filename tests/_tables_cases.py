"""Shared by tests/test_emu_tables.py (host SIMT emulator) and tests/test_gpu_tables.py (C-ABI hook on the GPU): histograms for
the stage tests of the wave-wide entropy-table builders, and the reference's own answers (oracle/_ref/libzref_shim.so)."""
import ctypes as C
import numpy as np

FSE_CT_DT = np.dtype([("state", "<u2", (512,)), ("dFind", "<i4", (56,)), ("dBits", "<u4", (56,)), ("tableLog", "<u4")])


def huf_cases(seed=0, n=160):
    """literal histograms: flat, skewed, many equal counts (tie order!), counts around the 165/166 bucket cut, > 11-bit trees"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = i % 8
        m = int(rng.integers(2, 256))
        c = np.zeros(256, dtype=np.uint32)
        if kind == 0:
            c[:m + 1] = rng.integers(0, 400, m + 1)
        elif kind == 1:
            c[:m + 1] = (rng.zipf(1.3, m + 1) % 5000).astype(np.uint32)
        elif kind == 2:
            c[:m + 1] = rng.integers(160, 172, m + 1)                   # around the exact-count / log2-bucket boundary
        elif kind == 3:
            c[:m + 1] = rng.choice([165, 165, 165, 300, 300, 700, 1], m + 1)   # many equal counts in sorted buckets
        elif kind == 4:
            c[:m + 1] = np.maximum(1, (2.0 ** (rng.random(m + 1) * 16)).astype(np.uint32))   # deep tree -> height limit
        elif kind == 5:
            f = [1, 1]
            while len(f) < min(m + 1, 30): f.append(f[-1] + f[-2])      # Fibonacci: the deepest possible tree
            c[:len(f)] = f
        elif kind == 6:
            c[:m + 1] = rng.integers(0, 3, m + 1) * rng.integers(1, 60000, m + 1)
        else:
            c[:m + 1] = rng.integers(1000, 1100, m + 1)
        nz = np.nonzero(c)[0]
        if len(nz) < 2:
            c[0] = 5; c[1] = 3; nz = np.nonzero(c)[0]
        out.append((c, int(nz[-1])))
    return out


def fse_cases(seed=0, n=240):
    """sequence-code histograms (<= 53 symbols) and weight histograms (13 symbols) with their table logs, both low-prob modes"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = i % 6
        alpha = int(rng.choice([13, 29, 32, 36, 53]))
        c = np.zeros(64, dtype=np.uint32)
        if kind == 0:
            c[:alpha] = rng.integers(0, 50, alpha)
        elif kind == 1:
            c[:alpha] = (rng.zipf(1.5, alpha) % 3000).astype(np.uint32)
        elif kind == 2:
            c[:alpha] = rng.integers(0, 2, alpha) * rng.integers(1, 4000, alpha)
        elif kind == 3:                                                     # one dominant symbol + many tiny ones: secondary distribution
            c[:alpha] = rng.integers(0, 3, alpha); c[int(rng.integers(0, alpha))] = int(rng.integers(2000, 30000))
        elif kind == 4:                                                     # near-uniform: everything close to one cell
            c[:alpha] = rng.integers(90, 110, alpha)
        else:
            c[:alpha] = rng.integers(0, 6, alpha) ** 3
        nz = np.nonzero(c)[0]
        if len(nz) < 2:
            c[0] += 7; c[1] += 2; nz = np.nonzero(c)[0]
        maxSym = int(nz[-1]); total = int(c.sum())
        if int(c.max()) == total:
            c[(int(nz[0]) + 1) % alpha] += 1; total += 1; maxSym = int(np.nonzero(c)[0][-1])
        maxLog = 6 if alpha == 13 else (8 if alpha in (29, 32) else 9)
        out.append((c, total, maxSym, maxLog, i % 2))
    return out


def ref_huf(lr, c, maxSym, maxNbBits=11):
    lr.zref_huf_write_table.restype = C.c_size_t
    lr.zref_huf_write_table.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    lr.zref_huf_build.restype = C.c_size_t
    lr.zref_huf_build.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
    nb = np.zeros(256, dtype=np.uint8)
    log = lr.zref_huf_build(c.ctypes.data_as(C.c_void_p), maxSym, maxNbBits, nb.ctypes.data_as(C.c_void_p))
    hdr = np.zeros(300, dtype=np.uint8); lo = C.c_uint(0)
    h = lr.zref_huf_write_table(hdr.ctypes.data_as(C.c_void_p), 300, c.ctypes.data_as(C.c_void_p), maxSym, maxNbBits, C.byref(lo))
    return int(log), nb, (None if h == C.c_size_t(-1).value else hdr[:h].tobytes())


def ref_fse(lr, c, total, maxSym, tableLog, lowProb):
    lr.zref_fse_normalize.restype = C.c_size_t
    lr.zref_fse_normalize.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_uint, C.c_uint]
    lr.zref_fse_write_ncount.restype = C.c_size_t
    lr.zref_fse_write_ncount.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.c_uint]
    lr.zref_fse_build_ctable.restype = C.c_size_t
    lr.zref_fse_build_ctable.argtypes = [C.c_void_p] * 4 + [C.c_uint, C.c_uint]
    norm = np.zeros(64, dtype=np.int16)
    r = lr.zref_fse_normalize(norm.ctypes.data_as(C.c_void_p), tableLog, c.ctypes.data_as(C.c_void_p), total, maxSym, lowProb)
    if r == C.c_size_t(-1).value:
        return -1, None, None, None
    if r == 0:
        return 0, None, None, None
    buf = np.zeros(600, dtype=np.uint8)
    h = lr.zref_fse_write_ncount(buf.ctypes.data_as(C.c_void_p), 600, norm.ctypes.data_as(C.c_void_p), maxSym, tableLog)
    st = np.zeros(1 << tableLog, dtype=np.uint16); df = np.zeros(64, dtype=np.int32); db = np.zeros(64, dtype=np.uint32)
    lr.zref_fse_build_ctable(st.ctypes.data_as(C.c_void_p), df.ctypes.data_as(C.c_void_p), db.ctypes.data_as(C.c_void_p), norm.ctypes.data_as(C.c_void_p), maxSym, tableLog)
    return 1, norm, buf[:h].tobytes(), (st, df, db)


def check_huf(run, lr, cases, maxNbBits=11):
    """run(counts[n,256] u32, maxSyms[n] u32, maxNbBits) -> codes[n,256] u32, hdrs[n,136] u8, meta[n,2] u32"""
    counts = np.stack([c for c, _ in cases]); maxSyms = np.array([m for _, m in cases], dtype=np.uint32)
    codes, hdrs, meta = run(counts, maxSyms, maxNbBits)
    for i, (c, m) in enumerate(cases):
        log, nb, hdr = ref_huf(lr, c, m, maxNbBits)
        got_nb = (codes[i] & 0xFF).astype(np.uint8)
        assert int(meta[i, 0]) == log and np.array_equal(got_nb[:m + 1], nb[:m + 1]), ("huffman lengths", i, m, log, int(meta[i, 0]))
        # canonical values: per length, counting up in symbol order from the reference's start value
        want_h = 0 if hdr is None else len(hdr)
        assert int(meta[i, 1]) == want_h, ("tree description size", i, int(meta[i, 1]), want_h)
        if hdr is not None:
            assert hdrs[i, :want_h].tobytes() == hdr, ("tree description bytes", i)


def check_fse(run, lr, cases):
    """run(counts[n,64] u32, params[n,4] u32) -> norms[n,64] i16, ncounts[n,64] u8, meta[n,2] i32, tables[n] FSE_CT_DT"""
    from_log = lambda total, maxSym, maxLog: lr.zref_fse_optimal_tablelog(maxLog, total, maxSym)
    lr.zref_fse_optimal_tablelog.restype = C.c_uint
    lr.zref_fse_optimal_tablelog.argtypes = [C.c_uint, C.c_size_t, C.c_uint]
    counts = np.stack([c for c, *_ in cases])
    params = np.array([[t, m, from_log(t, m, ml), lp] for _, t, m, ml, lp in cases], dtype=np.uint32)
    norms, ncounts, meta, tables = run(counts, params)
    seen_m2 = 0
    for i, (c, total, maxSym, maxLog, lp) in enumerate(cases):
        tl = int(params[i, 2])
        rc, norm, hdr, tab = ref_fse(lr, c, total, maxSym, tl, lp)
        assert int(meta[i, 0]) == rc, ("normalize rc", i, int(meta[i, 0]), rc)
        if rc != 1:
            continue
        assert np.array_equal(norms[i, :maxSym + 1], norm[:maxSym + 1]), ("norm", i, norms[i, :maxSym + 1].tolist(), norm[:maxSym + 1].tolist())
        assert int(meta[i, 1]) == len(hdr) and ncounts[i, :len(hdr)].tobytes() == hdr, ("NCount", i, int(meta[i, 1]), len(hdr))
        st, df, db = tab
        t = tables[i]
        assert int(t["tableLog"]) == tl and np.array_equal(t["state"][: 1 << tl], st), ("state table", i)
        assert np.array_equal(t["dBits"][: maxSym + 1], db[: maxSym + 1]), ("deltaNbBits", i)
        used = norm[: maxSym + 1] != 0
        assert np.array_equal(t["dFind"][: maxSym + 1][used], df[: maxSym + 1][used]), ("deltaFindState", i)
