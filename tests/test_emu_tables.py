"""Stage tests of the wave-wide entropy-table builders (zstd_amd/csrc/zhip_tables.h) on the host SIMT emulator, each pinned to
the REAL reference's stage function: HUF_buildCTable_wksp / HUF_writeCTable_wksp, FSE_normalizeCount / FSE_writeNCount /
FSE_buildCTable_wksp (oracle/_ref/libzref_shim.so).  CPU only."""
import ctypes as C
import numpy as np
import pytest
from _libs import load_emu, load_ref, have_ref
import _tables_cases as T

pytestmark = pytest.mark.skipif(not have_ref(), reason="needs oracle/_ref (the reference built from /root/reference)")


@pytest.fixture(scope="module")
def libs():
    return load_emu(), load_ref()


def emu_huf(le):
    def run(counts, maxSyms, maxNbBits):
        n = len(counts)
        codes = np.zeros((n, 256), dtype=np.uint32); hdrs = np.zeros((n, 136), dtype=np.uint8); meta = np.zeros((n, 2), dtype=np.uint32)
        le.emu_test_huf.restype = None
        le.emu_test_huf.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        le.emu_test_huf(np.ascontiguousarray(counts).ctypes.data_as(C.c_void_p), maxSyms.ctypes.data_as(C.c_void_p), n, maxNbBits,
                        codes.ctypes.data_as(C.c_void_p), hdrs.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p))
        return codes, hdrs, meta
    return run


def emu_fse(le):
    def run(counts, params):
        n = len(counts)
        assert le.emu_sizeof_fse_ctable() == T.FSE_CT_DT.itemsize
        norms = np.zeros((n, 64), dtype=np.int16); ncounts = np.zeros((n, 64), dtype=np.uint8); meta = np.zeros((n, 2), dtype=np.int32)
        tables = np.zeros(n, dtype=T.FSE_CT_DT)
        le.emu_test_fse.restype = None
        le.emu_test_fse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint] + [C.c_void_p] * 4
        le.emu_test_fse(np.ascontiguousarray(counts).ctypes.data_as(C.c_void_p), np.ascontiguousarray(params).ctypes.data_as(C.c_void_p), n,
                        norms.ctypes.data_as(C.c_void_p), ncounts.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p), tables.ctypes.data_as(C.c_void_p))
        return norms, ncounts, meta, tables
    return run


def test_huffman_code_and_tree_description_match_reference(libs):
    le, lr = libs
    T.check_huf(emu_huf(le), lr, T.huf_cases(seed=1, n=160))


def test_huffman_lower_height_limits(libs):
    le, lr = libs
    for lim in (8, 9, 10):
        T.check_huf(emu_huf(le), lr, T.huf_cases(seed=10 + lim, n=48), maxNbBits=lim)


def test_fse_normalize_ncount_ctable_match_reference(libs):
    le, lr = libs
    T.check_fse(emu_fse(le), lr, T.fse_cases(seed=2, n=240))
