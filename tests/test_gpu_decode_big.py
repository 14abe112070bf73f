"""-m gpu: ONE large frame decoded block-parallel (zstd_amd/csrc/zhip_decode_big.h) — the frames this library's single-frame and
job-pool modes emit (and any other frame that states its content size and needs no dictionary).  Same bytes as the per-frame
decoder, same error codes (whatever the block-parallel path declines goes through k_decode), content checksums verified.
The path itself ran on MI355X (profiles/r03_big_frame_decode_*: the 1 GiB frame, bit-exact); this file sorts last in the suite because
its own first complete GPU run is the driver's (the round's GPU budget ended before it)."""
import ctypes as C
import numpy as np
import pytest
from _libs import load_oracle, oracle_frame, oracle_frame_mt, oracle_frame_params, datagen, text_like, _buf

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import zstd_amd
    zstd_amd.lib()
    return zstd_amd, load_oracle()


def test_job_pool_frame_is_decoded_block_parallel(env):
    z, lo = env
    a = z.datagen(96 << 20, 50, 3)
    ctx = z.Context(max_units=256)
    for level, ck in ((1, False), (3, True)):
        ctx.set_checksum(ck)
        frame = ctx.compress_frames([a], level, workers=2)[0]
        d = z.DContext()
        out = d.decompress(frame)
        info = d.last_bigframe()
        assert out == a.tobytes(), (level, ck)
        assert info["block_parallel"] == 1 and info["fell_back"] == 0 and info["blocks"] >= (96 << 20) // (128 << 10), info
        d.set_bigframe_min(0)                                    # the same frame through one workgroup: same bytes
        assert d.decompress(frame[:]) == out and d.last_bigframe()["block_parallel"] == 0
    # device-resident frame (zhip_decompress_frames_device): the block headers are walked on the device (k_bf_walk) instead of on the host
    import torch
    d = z.DContext()
    blob = np.frombuffer(frame + b"\x00" * 64, dtype=np.uint8)
    srcT = torch.from_numpy(blob.copy()).cuda()
    outT = torch.zeros(len(a) + 64, dtype=torch.uint8, device="cuda")
    r, status, dsz = d.decompress_frames_device(outT.data_ptr(), np.array([0], np.uint64), np.array([len(a)], np.uint64), srcT.data_ptr(),
                                                np.array([0], np.uint64), np.array([len(frame)], np.uint64))
    assert status[0] == 0 and int(dsz[0]) == len(a) and d.last_bigframe()["block_parallel"] == 1
    assert outT[:len(a)].cpu().numpy().tobytes() == a.tobytes()
    del srcT, outT
    ctx.set_checksum(True)
    frame = bytearray(ctx.compress_frames([a[:20 << 20]], 1, workers=2)[0])
    frame[-1] ^= 0x55                                            # a wrong content checksum is still found
    d = z.DContext()
    with pytest.raises(z.ZhipError):
        d.decompress(bytes(frame))


def test_big_frames_of_every_strategy_and_shape(env):
    z, lo = env
    rng = np.random.default_rng(3)
    mixed = np.concatenate([datagen(lo, 3 << 20, 50, 1), rng.integers(0, 256, size=2 << 20, dtype=np.uint8), np.full(3 << 20, 7, np.uint8),
                            text_like(3 << 20, 9), np.tile(rng.integers(0, 256, size=700, dtype=np.uint8), 3000)])
    ctx = z.Context(max_units=64)
    d = z.DContext()
    d.set_bigframe_min(1 << 20)
    for level in (1, 3, -1):
        frame = ctx.compress_frames([mixed], level)[0]           # one multi-block frame: treeless literals, RLE / raw blocks, long periodic matches
        assert d.decompress(frame) == mixed.tobytes(), level
        assert d.last_bigframe()["block_parallel"] == 1, (level, d.last_bigframe())
    light = np.concatenate([text_like(2 << 20, 4), datagen(lo, 256 << 10, 50, 6)])      # the lazy strategies repeat FSE tables by cost (their frame kernel is slow on long matches: kept small)
    for level, row in ((5, 0), (8, 2)):
        ctx.set_row_matcher(row)
        frame = ctx.compress_frames([light], level)[0]
        assert d.decompress(frame) == light.tobytes(), level
        assert d.last_bigframe()["block_parallel"] == 1, (level, d.last_bigframe())
    # several frames in one call, large and small mixed: the large ones block-parallel, the others as a batch, results in order
    small = [datagen(lo, n, 50, n) for n in (0, 5, 70000, 300000)]
    frames = [ctx.compress_frames([b], 1)[0] for b in small]
    big1 = ctx.compress_frames([mixed], 1)[0]
    blob = frames[0] + big1 + frames[1] + frames[2] + big1 + frames[3]
    want = small[0].tobytes() + mixed.tobytes() + small[1].tobytes() + small[2].tobytes() + mixed.tobytes() + small[3].tobytes()
    assert d.decompress(blob) == want
    assert d.last_bigframe()["block_parallel"] == 2


def test_damaged_big_frames_report_the_per_frame_decoders_errors(env):
    z, lo = env
    a = z.datagen(6 << 20, 50, 5)
    ctx = z.Context(max_units=64)
    base = ctx.compress_frames([a], 3)[0]
    rng = np.random.default_rng(17)
    dbig, dser = z.DContext(), z.DContext()
    dbig.set_bigframe_min(1 << 20); dser.set_bigframe_min(0)
    differ = 0
    for trial in range(24):
        f = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(6, len(f)))
            f[p] ^= 1 << int(rng.integers(0, 8))
        res = []
        for d in (dbig, dser):
            try:
                res.append(("ok", d.decompress(bytes(f), capacity=len(a))))
            except z.ZhipError as e:
                res.append(("err", str(e).split("(")[0]))
        assert res[0] == res[1], trial
        differ += res[0][0] == "err"
    assert differ > 0
