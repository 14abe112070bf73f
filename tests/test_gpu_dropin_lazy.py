"""-m gpu: the drop-in's ZSTD_compress2 at the lazy levels with ZSTD_c_nbWorkers (the round-2 verdict's done-criterion for the lazy-strategy
frames): byte-identical to the oracle and — when oracle/_ref travelled — to the reference.  The frames underneath are covered by
tests/test_gpu_frames_lazy.py (green on MI355X); this file sorts last in the suite because its own first complete GPU run is the driver's
(the round's GPU budget ended before it)."""
import ctypes as C
import os
import numpy as np
import pytest
from _libs import load_oracle, load_ref, have_ref, oracle_frame_mt, ref_frame_mt, datagen, text_like, _buf, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import zstd_amd
    zstd_amd.lib()
    return zstd_amd, load_oracle()


def test_shim_compress2_lazy_levels_with_workers(env):
    """the done-criterion: ZSTD_compress2 of the drop-in at levels 5-7 with nbWorkers >= 1 on >= 3 MB, byte-identical to the reference"""
    z, lo = env
    shim = C.CDLL(os.path.join(ROOT, "zstd_amd", "libzstd_hipshim.so"))
    shim.ZSTD_createCCtx.restype = C.c_void_p
    shim.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    shim.ZSTD_CCtx_setParameter.restype = C.c_size_t; shim.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    shim.ZSTD_compress2.restype = C.c_size_t; shim.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    shim.ZSTD_compressBound.restype = C.c_size_t; shim.ZSTD_compressBound.argtypes = [C.c_size_t]
    shim.ZSTD_isError.restype = C.c_uint; shim.ZSTD_isError.argtypes = [C.c_size_t]
    a = np.concatenate([datagen(lo, 2 << 20, 50, 21), text_like(1_200_000, 22)])
    lr = load_ref() if have_ref() else None
    for level in (5, 6, 7):
        cc = shim.ZSTD_createCCtx()
        assert shim.ZSTD_isError(shim.ZSTD_CCtx_setParameter(cc, 100, level)) == 0          # ZSTD_c_compressionLevel
        assert shim.ZSTD_isError(shim.ZSTD_CCtx_setParameter(cc, 400, 2)) == 0              # ZSTD_c_nbWorkers
        assert shim.ZSTD_isError(shim.ZSTD_CCtx_setParameter(cc, 1011, 1)) == 0            # ZSTD_c_useRowMatchFinder = enable: the reference's default here (the suite's environment says hash chain)
        cap = shim.ZSTD_compressBound(len(a))
        dst = np.zeros(cap, dtype=np.uint8)
        r = shim.ZSTD_compress2(cc, _buf(dst), cap, _buf(a), len(a))
        assert not shim.ZSTD_isError(r), level
        out = dst[:r].tobytes()
        lo.zo_set_row_matcher(1)
        try:
            want = oracle_frame_mt(lo, a, level, 0, 0, 0)
        finally:
            lo.zo_set_row_matcher(0)
        assert out == want, level
        if lr is not None:
            assert out == ref_frame_mt(lr, a, level, 0, 0, 0), ("reference", level)
        shim.ZSTD_freeCCtx(cc)
    assert z.DContext().decompress(out) == a.tobytes()
