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
