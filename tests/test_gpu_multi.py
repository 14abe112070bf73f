"""In-process multi-device host path (zhip_compress_multi): lanes on separate streams, pinned double buffers, ordered
host gather.  One GPU is visible here, so the device list names it several times — the sharding, the lane threads and the gather
are the same code that runs over 8 devices; the stream must be byte-identical to the single-context path and to the oracle."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest
from _libs import load_oracle, datagen, text_like, _buf, ERR, ROOT

pytestmark = pytest.mark.gpu
UNIT = 131072


def test_multi_lane_stream_equals_single_context_and_oracle():
    import zstd_amd
    lo = load_oracle()
    a = np.concatenate([datagen(lo, 37 * UNIT + 777, 50, 4), text_like(11 * UNIT, 9)])
    single = zstd_amd.Context(0, max_units=64).compress(a, level=1)
    cap = lo.zo_compress_bound(UNIT) * 50
    want = np.empty(cap, dtype=np.uint8)
    r = lo.zo_compress_chunks(1, UNIT, _buf(a), len(a), _buf(want), cap, None, 0)
    assert r != ERR and single == want[:r].tobytes()
    for devices, chunk in (([0], 0), ([0, 0], 3), ([0, 0, 0], 1), ([0], 5)):          # 2..6 lanes, chunks of 1..256 units: many gather orders
        m = zstd_amd.MultiContext(devices, chunk_units=chunk)
        sizes = np.zeros(64, dtype=np.uint64)
        dst = np.empty(zstd_amd.compress_bound(len(a)), dtype=np.uint8)
        for rep in range(3):
            k = m.compress_into(dst, a, level=1, sizes=sizes)
            assert dst[:k].tobytes() == single, (devices, chunk, rep)
        assert int(sizes[:49].sum()) == k
        got3 = m.compress(a[: 5 * UNIT + 1], level=3)
        assert got3 == zstd_amd.Context(0, max_units=8).compress(a[: 5 * UNIT + 1], level=3)
        assert m.compress(np.zeros(0, dtype=np.uint8), level=1) == zstd_amd.Context(0, max_units=1).compress(np.zeros(0, dtype=np.uint8), level=1)
        m.close()


def test_shim_uses_the_multi_device_path_when_asked(tmp_path):
    """ZHIP_DEVICES=0,0: ZSTD_compress2 of a 3 MiB source through libzstd_hipshim.so runs on the lanes of zhip_compress_multi"""
    import zstd_amd
    from zstd_amd import build as zbuild
    lo = load_oracle()
    a = datagen(lo, 24 * UNIT + 5, 50, 8)
    want = zstd_amd.Context(0, max_units=32).compress(a, level=1)
    code = f'''
import ctypes as C, numpy as np, sys
S = C.CDLL({zbuild.SHIM!r})
S.ZSTD_createCCtx.restype = C.c_void_p
S.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
S.ZSTD_compress2.restype = C.c_size_t; S.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
S.ZSTD_compressBound.restype = C.c_size_t; S.ZSTD_compressBound.argtypes = [C.c_size_t]
a = np.fromfile({str(tmp_path / "in.bin")!r}, dtype=np.uint8)
c = S.ZSTD_createCCtx(); S.ZSTD_CCtx_setParameter(c, 100, 1)
cap = S.ZSTD_compressBound(len(a)); dst = np.zeros(cap, dtype=np.uint8)
r = S.ZSTD_compress2(c, dst.ctypes.data_as(C.c_void_p), cap, a.ctypes.data_as(C.c_void_p), len(a))
dst[:r].tofile({str(tmp_path / "out.bin")!r})
'''
    a.tofile(tmp_path / "in.bin")
    env = dict(os.environ, ZHIP_DEVICES="0,0")
    subprocess.check_call([os.sys.executable, "-c", code], env=env)
    assert open(tmp_path / "out.bin", "rb").read() == want


def test_job_pool_frame_over_lanes_equals_the_single_context_frame():
    """zhip_compress_frame_mt_multi: ONE frame (ZSTD_c_nbWorkers semantics), chunks of consecutive jobs on different lanes / devices,
    ordered gather — the bytes of zhip_compress_frames_mt on one context and of the oracle, whatever the chunking"""
    import zstd_amd
    from _libs import oracle_frame_mt, oracle_frame
    lo = load_oracle()
    os.environ["ZHIP_MULTI_FRAME_CHUNKED"] = "1"                                    # one GPU here: force the several-device path
    a = np.concatenate([datagen(lo, 5_000_000, 50, 4), text_like(1_300_000, 9), np.zeros(700_000, np.uint8)])
    for level, js, ov in ((1, 524288, 0), (3, 524288, 0), (1, 0, 9), (-1, 600_001, 3)):
        want = oracle_frame_mt(lo, a, level, js, ov)
        for devices, chunk in (([0], 0), ([0, 0], 8), ([0], 5)):                     # 64 MB lanes (one chunk), 1 MiB and 640 KB staging: one job per chunk
            m = zstd_amd.MultiContext(devices, chunk_units=chunk)
            for rep in range(2):
                assert m.compress_frame_mt(a, level, job_size=js, overlap_log=ov) == want, (level, js, ov, devices, chunk, rep)
            m.close()
    m = zstd_amd.MultiContext([0, 0], chunk_units=16)
    small = a[:400_000]
    assert m.compress_frame_mt(small, 1) == oracle_frame(lo, small, 1)               # at or below 512 KB: the plain frame
    m.set_checksum(True)
    got = m.compress_frame_mt(a, 1, job_size=524288)                                 # checksum: one lane's zhip_compress_frames_mt
    assert got == oracle_frame_mt(lo, a, 1, 524288, 0, True)
    with pytest.raises(zstd_amd.ZhipError):
        m.compress_frame_mt(a, 13)                                                   # btlazy2: not a device strategy, no CPU fallback
    m.close()
    del os.environ["ZHIP_MULTI_FRAME_CHUNKED"]
    m = zstd_amd.MultiContext([0])                                                   # one device: the single-context path
    assert m.compress_frame_mt(a, 1) == oracle_frame_mt(lo, a, 1)
    m.close()


def test_multi_create_rejects_devices_that_do_not_exist():
    """two lanes on device ordinals the box does not have: the constructor fails cleanly (NULL -> ZhipError), nothing is left behind,
    and a valid context still works afterwards"""
    import torch
    import zstd_amd
    assert torch.cuda.is_available()
    n = zstd_amd.lib().zhip_device_count()
    with pytest.raises(zstd_amd.ZhipError):
        zstd_amd.MultiContext([n + 3, n + 4])
    with pytest.raises(zstd_amd.ZhipError):
        zstd_amd.MultiContext([0, n + 1])
    m = zstd_amd.MultiContext([0])
    a = np.arange(300000, dtype=np.uint32).view(np.uint8)
    assert len(m.compress(a, level=1)) > 0
    m.close()


def test_two_ranks_on_one_device_through_bench_py():
    """The N > 1 path with the REAL per-rank compressor: `bench.py --gpus 2` starts two ranks (torch.distributed.run, gloo control plane) that share
    device 0, each compresses its own datagen shard (the weak-scaling line) and its shard of ONE text buffer (the frame-per-shard form of BASELINE
    configs[3]) and copies its frames to its offset of one shared host buffer.  The gathered stream must be the stream a single process makes of
    the whole buffer, and every rank's stream the real reference's."""
    import hashlib
    import json
    import sys
    import zstd_amd
    from zstd_amd import workloads as W
    total = 24 * UNIT + 12345
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--mib", "8", "--no-cpu-baseline",
                         "--total-bytes", str(total)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    assert cp.returncode == 0 and lines, cp.stderr[-2000:]
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["parity"]["bytes_identical_to_oracle_first_64_units"] and out["parity"]["frames_well_formed"]
    if "full_size" in out["parity"]:
        assert out["parity"]["full_size"]["sha256_equals_reference_stream"]
    ts = out["text_strong_scaling"]
    assert ts["n_gpus"] == 2 and ts["scaling"] == "strong"
    assert list(out.keys())[-1] == "digest" and "text1e9_strong" in out["digest"]
    a = W.tile(W.text_corpus(64 << 20, seed=0), total)
    single = zstd_amd.Context(0, max_units=32).compress(a, level=1)
    assert ts["parity"]["gathered_bytes"] == len(single)
    assert ts["parity"]["gathered_stream_sha256"] == hashlib.sha256(single).hexdigest()
    assert all(x in (True, None) for x in ts["parity"]["per_rank_sha256_equals_reference"])


def test_eight_ranks_on_one_device_through_bench_py():
    """The first 8-GPU lease must not fail on plumbing (round-5 verdict, item 7): `bench.py --gpus 8` with eight real ranks sharing device 0 on a small --mib —
    one JSON line, n_gpus 8, the weak-scaling parity of rank 0, the frame-per-shard text leg with eight shards gathered into ONE stream that equals a single
    process's, every rank's stream the real reference's."""
    import hashlib
    import json
    import sys
    import zstd_amd
    from zstd_amd import workloads as W
    total = 41 * UNIT + 777                                       # 42 units over 8 ranks: six per rank, the last rank's range is short (and ragged)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--mib", "4", "--no-cpu-baseline",
                         "--total-bytes", str(total)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    lines = [l for l in cp.stdout.splitlines() if l.startswith("{")]
    assert cp.returncode == 0 and lines, cp.stderr[-3000:]
    assert len(lines) == 1, "rank 0 prints ONE line"
    out = json.loads(lines[-1])
    assert out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["units_per_gpu"] == 32
    assert out["parity"]["bytes_identical_to_oracle_first_64_units"] and out["parity"]["frames_well_formed"]
    ts = out["text_strong_scaling"]
    assert ts["n_gpus"] == 8 and ts["scaling"] == "strong"
    assert list(out.keys())[-1] == "digest" and "text1e9_strong" in out["digest"]
    a = W.tile(W.text_corpus(64 << 20, seed=0), total)
    single = zstd_amd.Context(0, max_units=64).compress(a, level=1)
    assert ts["parity"]["gathered_bytes"] == len(single)
    assert ts["parity"]["gathered_stream_sha256"] == hashlib.sha256(single).hexdigest()
    assert len(ts["parity"]["per_rank_sha256_equals_reference"]) == 8
    assert all(x in (True, None) for x in ts["parity"]["per_rank_sha256_equals_reference"])


def test_multi_context_with_eight_devices_entries_on_one_gpu():
    """zhip_compress_multi with devices = {0,0,0,0,0,0,0,0}: eight "devices" (sixteen lanes) on the one GPU of the box — the lane / ordered-gather machinery at the
    width an 8-GPU node gives it; the stream equals a single context's and the oracle's"""
    import zstd_amd
    lo = load_oracle()
    a = np.concatenate([datagen(lo, 40 * UNIT + 999, 50, 11), text_like(9 * UNIT + 17, 5)])
    want = zstd_amd.Context(0, max_units=64).compress(a, level=1)
    m = zstd_amd.MultiContext([0] * 8, chunk_units=2)              # 256 KB chunks: 25 chunks over 16 lanes
    for rep in range(2):
        assert m.compress(a, level=1) == want, rep
    assert m.compress(a, level=3) == zstd_amd.Context(0, max_units=64).compress(a, level=3)
    m.close()
    m = zstd_amd.MultiContext([0] * 8)                             # default chunking: one or two chunks, most lanes idle
    assert m.compress(a, level=1) == want
    m.close()
