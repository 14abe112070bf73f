"""Stage tests of the wave-wide entropy-table builders ON THE GPU, through the C-ABI test hooks zhip_test_huf_tables /
zhip_test_fse_tables: the same histograms as tests/test_emu_tables.py.  Where oracle/_ref travels (it does: a prebuilt binary) the
answers come from the REAL reference's stage functions; the committed golden digests (tests/golden/tables_v1.json, made by
tests/golden/make_tables_golden.py from the reference) pin them on a box without it."""
import ctypes as C
import hashlib
import json
import os
import numpy as np
import pytest
from _libs import load_ref, have_ref, ROOT
import _tables_cases as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import zstd_amd
    return zstd_amd, zstd_amd.Context(0, max_units=8)


def gpu_huf(zstd_amd, ctx):
    L = zstd_amd.lib()
    L.zhip_test_huf_tables.restype = C.c_size_t
    L.zhip_test_huf_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]

    def run(counts, maxSyms, maxNbBits):
        n = len(counts)
        codes = np.zeros((n, 256), dtype=np.uint32); hdrs = np.zeros((n, 136), dtype=np.uint8); meta = np.zeros((n, 2), dtype=np.uint32)
        r = L.zhip_test_huf_tables(ctx._h, np.ascontiguousarray(counts).ctypes.data_as(C.c_void_p), maxSyms.ctypes.data_as(C.c_void_p), n, maxNbBits,
                                   codes.ctypes.data_as(C.c_void_p), hdrs.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p))
        assert r == 0
        return codes, hdrs, meta
    return run


def gpu_fse(zstd_amd, ctx):
    L = zstd_amd.lib()
    L.zhip_test_fse_tables.restype = C.c_size_t
    L.zhip_test_fse_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint] + [C.c_void_p] * 4 + [C.c_size_t]

    def run(counts, params):
        n = len(counts)
        norms = np.zeros((n, 64), dtype=np.int16); ncounts = np.zeros((n, 64), dtype=np.uint8); meta = np.zeros((n, 2), dtype=np.int32)
        tables = np.zeros(n, dtype=T.FSE_CT_DT)
        r = L.zhip_test_fse_tables(ctx._h, np.ascontiguousarray(counts).ctypes.data_as(C.c_void_p), np.ascontiguousarray(params).ctypes.data_as(C.c_void_p), n,
                                   norms.ctypes.data_as(C.c_void_p), ncounts.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p),
                                   tables.ctypes.data_as(C.c_void_p), T.FSE_CT_DT.itemsize)
        assert r == 0
        return norms, ncounts, meta, tables
    return run


def test_gpu_huffman_tables_match_reference(ctx):
    zstd_amd, c = ctx
    if have_ref():
        T.check_huf(gpu_huf(zstd_amd, c), load_ref(), T.huf_cases(seed=1, n=160))
        T.check_huf(gpu_huf(zstd_amd, c), load_ref(), T.huf_cases(seed=18, n=48), maxNbBits=8)
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "tables_v1.json")))
    cases = T.huf_cases(seed=1, n=160)
    codes, hdrs, meta = gpu_huf(zstd_amd, c)(np.stack([x for x, _ in cases]), np.array([m for _, m in cases], dtype=np.uint32), 11)
    h = hashlib.sha256()
    for i, (_, m) in enumerate(cases):
        h.update((codes[i] & 0xFF).astype(np.uint8)[: m + 1].tobytes()); h.update(bytes([int(meta[i, 0]), int(meta[i, 1])])); h.update(hdrs[i, : int(meta[i, 1])].tobytes())
    assert h.hexdigest() == gold["huf_seed1_n160_max11"]


def test_gpu_fse_tables_match_reference(ctx):
    zstd_amd, c = ctx
    if have_ref():
        T.check_fse(gpu_fse(zstd_amd, c), load_ref(), T.fse_cases(seed=2, n=240))
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "tables_v1.json")))
    cases = T.fse_cases(seed=2, n=240)
    counts = np.stack([x for x, *_ in cases])
    params = np.array(gold["fse_seed2_n240_params"], dtype=np.uint32)
    norms, ncounts, meta, tables = gpu_fse(zstd_amd, c)(counts, params)
    h = hashlib.sha256()
    for i, (_, total, maxSym, _, _) in enumerate(cases):
        h.update(meta[i].tobytes())
        if meta[i, 0] == 1:
            tl = int(params[i, 2])
            h.update(norms[i, : maxSym + 1].tobytes()); h.update(ncounts[i, : int(meta[i, 1])].tobytes())
            h.update(tables[i]["state"][: 1 << tl].tobytes()); h.update(tables[i]["dBits"][: maxSym + 1].tobytes())
    assert h.hexdigest() == gold["fse_seed2_n240"]
