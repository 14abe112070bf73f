"""-m gpu: structured-random parity fuzz through the C ABI on the real device: many small independent units per call
(one frame per `unit_size` chunk), every level the device implements, frames byte-identical to the oracle."""
import numpy as np
import pytest
from _libs import load_oracle, _buf, ERR
from test_fuzz_emu import gen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("level", [-5, -1, 1, 2, 3, 4, 5, 6, 7, 9])
def test_fuzz_units_through_the_c_abi(level):
    import torch
    import zstd_amd
    assert torch.cuda.is_available()
    lo = load_oracle()
    ctx = zstd_amd.Context(0, max_units=4096)
    dctx = zstd_amd.DContext(0)
    for unit, count, seed in ((700, 900, 1), (3000, 500, 2), (17000, 120, 3), (70000, 24, 4)):
        try:
            zstd_amd.get_cparams(level, unit)
        except zstd_amd.ZhipError:
            continue
        rng = np.random.default_rng(seed * 1000 + abs(level))
        tail = int(rng.integers(0, unit))
        try:
            zstd_amd.get_cparams(level, tail)
        except zstd_amd.ZhipError:
            tail = 0                                 # e.g. level 9 below 16 KB is btlazy2: keep the call to implemented strategies
        a = np.concatenate([gen(rng, unit) for _ in range(count)] + [gen(rng, tail)])
        got, sizes = ctx.compress(a, level=level, unit_size=unit, return_sizes=True)
        cap = lo.zo_compress_bound(unit) * (count + 2)
        dst = np.zeros(cap, dtype=np.uint8)
        nun = max(1, -(-len(a) // unit))
        osz = np.zeros(nun, dtype=np.uint64)
        r = lo.zo_compress_chunks(level, unit, _buf(a), len(a), _buf(dst), cap, _buf(osz), nun)
        assert r != ERR
        if got != dst[:r].tobytes():
            bad = [i for i in range(len(osz)) if int(sizes[i]) != int(osz[i])]
            raise AssertionError(f"level {level} unit {unit}: frames differ; first size mismatch at units {bad[:5]}")
        assert dctx.decompress(got, capacity=len(a)) == a.tobytes(), f"level {level} unit {unit}: the device decoder does not return the source"
    ctx.close(); dctx.close()


def test_fuzz_explicit_parameters_slice():
    """A bounded slice (about 200 units) of tests/tools/gpu_fuzz_units.py in the driver's suite: random unit sizes x random explicit
    parameters (every strategy up to lazy2, row matcher on / off) through zhip_compress_params, unit by unit against the oracle, and back
    through the device decoder."""
    import ctypes as C
    import torch
    import zstd_amd as z
    from _libs import datagen
    from test_fuzz_emu import _explicit
    assert torch.cuda.is_available()
    lo = load_oracle()
    lo.zo_compress_unit_params.restype = C.c_size_t
    lo.zo_compress_unit_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    L = z.lib()
    L.zhip_compress_params.restype = C.c_size_t
    L.zhip_compress_params.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(20260924)
    ctx = z.Context(max_units=256)
    dctx = z.DContext()
    units = cases = 0
    try:
        for t in range(60):
            if units >= 200:
                break
            unit = int(rng.choice([131072, 65536, 20000, 4096, 100000]))
            n = int(rng.integers(unit, 8 * unit))
            a = gen(rng, n) if t % 3 else np.concatenate([gen(rng, n // 2), datagen(lo, n - n // 2, int(rng.integers(5, 95)), t)])
            level = int(rng.choice([1, 3, 5, 6, 7, -3]))
            req = [int(rng.choice([0, 0, 17, 18])), int(rng.choice([0, 0, 8, 12, 15, 16])), int(rng.choice([0, 0, 8, 11, 13, 15, 17, 18])),
                   int(rng.choice([0, 0, 1, 2, 4, 5, 6])), int(rng.choice([0, 0, 3, 4, 5, 6, 7])), int(rng.choice([0, 0, 1, 4, 16, 64])), int(rng.choice([0, 0, 1, 2, 3, 4, 5]))]
            no_row = int(rng.integers(0, 2))
            ctx.set_row_matcher(2 if no_row else 0)
            cap = z.compress_bound(n, unit)
            dst = np.empty(cap, dtype=np.uint8); sizes = np.zeros(n // unit + 2, dtype=np.uint64)
            r = L.zhip_compress_params(ctx._h, dst.ctypes.data_as(C.c_void_p), cap, a.ctypes.data_as(C.c_void_p), n, level, (C.c_uint * 7)(*req), unit,
                                       sizes.ctypes.data_as(C.c_void_p))
            if L.zhip_isError(r):
                continue                                     # parameters the device does not run
            cases += 1
            assert dctx.decompress(dst[:r].tobytes()) == a.tobytes(), ("device decoder", t, level, req)
            pos = 0
            for k in range(-(-n // unit)):
                u = a[k * unit: (k + 1) * unit]
                eff = _explicit(level, len(u), req)
                lo.zo_set_row_matcher(1 if (3 <= eff[6] <= 5 and eff[0] > 14 and not no_row) else 0)
                o = np.zeros(lo.zo_compress_bound(len(u)) + 64, dtype=np.uint8)
                rr = lo.zo_compress_unit_params(_buf(o), len(o), _buf(u), len(u), eff)
                got = dst[pos: pos + int(sizes[k])].tobytes(); pos += int(sizes[k])
                units += 1
                assert rr != ERR and got == o[:rr].tobytes(), ("unit", t, k, len(u), level, req, list(eff), no_row)
    finally:
        lo.zo_set_row_matcher(0)
        ctx.close(); dctx.close()
    assert cases >= 10 and units >= 100


def test_lazy_units_at_hashlog_18_the_chain_builder_asks_for_more_than_64_KB_of_lds():
    """hashLog 18 is the largest the reference's adjustment leaves a 128 KB unit (windowLog + 1, zstd_compress.c:1466): the row matcher's builder then needs 69 952 bytes
    of LDS (hc_chain_lds_bytes), more than a launch gets without hipFuncAttributeMaxDynamicSharedMemorySize (round-5 advisor finding: the attribute was missing and
    no test reached the case).  Greedy / lazy / lazy2 with explicit parameters, row matcher on and off, unit by unit against the oracle."""
    import ctypes as C
    import torch
    import zstd_amd as z
    from _libs import datagen
    from test_fuzz_emu import _explicit
    assert torch.cuda.is_available()
    lo = load_oracle()
    lo.zo_compress_unit_params.restype = C.c_size_t
    lo.zo_compress_unit_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    L = z.lib()
    L.zhip_compress_params.restype = C.c_size_t
    L.zhip_compress_params.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(18)
    unit = 131072
    n = 3 * unit + 7777
    a = np.concatenate([gen(rng, n // 2), datagen(lo, n - n // 2, 40, 3)])
    ctx = z.Context(max_units=8)
    dctx = z.DContext()
    ran = 0
    try:
        for strat in (3, 4, 5):
            for no_row in (0, 1):
                req = [18, 16, 18, 4, 5, 16, strat]
                ctx.set_row_matcher(2 if no_row else 0)
                cap = z.compress_bound(n, unit)
                dst = np.empty(cap, dtype=np.uint8); sizes = np.zeros(n // unit + 2, dtype=np.uint64)
                r = L.zhip_compress_params(ctx._h, dst.ctypes.data_as(C.c_void_p), cap, a.ctypes.data_as(C.c_void_p), n, 5, (C.c_uint * 7)(*req), unit, sizes.ctypes.data_as(C.c_void_p))
                assert not L.zhip_isError(r), (strat, no_row, z.lib().zhip_getErrorName(r))
                assert dctx.decompress(dst[:r].tobytes()) == a.tobytes()
                pos = 0
                for k in range(-(-n // unit)):
                    u = a[k * unit: (k + 1) * unit]
                    eff = _explicit(5, len(u), req)
                    if len(u) == unit:
                        assert eff[2] == 18, list(eff)                     # the case this test exists for
                    lo.zo_set_row_matcher(1 if (3 <= eff[6] <= 5 and eff[0] > 14 and not no_row) else 0)
                    o = np.zeros(lo.zo_compress_bound(len(u)) + 64, dtype=np.uint8)
                    rr = lo.zo_compress_unit_params(_buf(o), len(o), _buf(u), len(u), eff)
                    got = dst[pos: pos + int(sizes[k])].tobytes(); pos += int(sizes[k])
                    assert rr != ERR and got == o[:rr].tobytes(), ("unit", strat, no_row, k, list(eff))
                    ran += 1
    finally:
        lo.zo_set_row_matcher(0)
        ctx.close(); dctx.close()
    assert ran == 24


def test_fuzz_frames_and_job_frames_slice():
    """A bounded slice (20 cases) of tests/tools/gpu_fuzz_frames.py: random inputs x explicit parameters x job sizes x overlaps x checksum
    through zhip_compress_frames / zhip_compress_frames_mt against the oracle (which the CPU suite pins to the reference on the same
    generator)."""
    import ctypes as C
    import torch
    import zstd_amd as z
    from _libs import oracle_frame_mt
    from test_oracle_vs_reference import mt_explicit_cases
    assert torch.cuda.is_available()
    lo = load_oracle()
    lo.zo_compress_frame_params.restype = C.c_size_t
    lo.zo_compress_frame_params.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    lo.zo_frame_bound.restype = C.c_size_t
    lo.zo_frame_bound.argtypes = [C.c_size_t]
    ctx = z.Context(max_units=32)
    n = 0
    for a, level, req, eff, js, ov, ck in mt_explicit_cases(lo, 20, 77):
        ctx.set_checksum(ck)
        got = ctx.compress_frames([a], level, cparams=req, workers=1, job_size=js, overlap_log=ov)[0]
        assert got == oracle_frame_mt(lo, a, level, js, ov, ck, cp=eff), ("job frame", len(a), level, req, js, ov, ck)
        ctx.set_checksum(False)
        cap = lo.zo_frame_bound(len(a)); o = np.zeros(cap, dtype=np.uint8)
        r = lo.zo_compress_frame_params(_buf(o), cap, _buf(a), len(a), eff)
        assert ctx.compress_frames([a], level, cparams=req)[0] == o[:r].tobytes(), ("frame", len(a), level, req)
        n += 1
    ctx.close()
    assert n >= 10
