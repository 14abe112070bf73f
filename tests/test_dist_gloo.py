"""world_size-2 gloo test of the N>1 path (zstd_amd/shard.py): contiguous unit ranges per rank, no collective in the
data path, ordered host gather.  The per-rank compressor is a test double (the oracle) because no GPU exists here;
the sharding / gather logic under test is the product's."""
import os
import sys
import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from _libs import load_oracle, datagen, _buf, ERR
    from zstd_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo = load_oracle()
    data = datagen(lo, n, 50, 5)

    def oracle_compress(buf, level, unit):
        buf = np.ascontiguousarray(buf)
        nu = max(1, -(-len(buf) // unit))
        cap = lo.zo_compress_bound(unit) * nu + 64
        dst = np.zeros(cap, dtype=np.uint8); sizes = np.zeros(nu, dtype=np.uint64)
        r = lo.zo_compress_chunks(level, unit, _buf(buf), len(buf), _buf(dst), cap, _buf(sizes), nu)
        assert r != ERR
        return dst[:r].tobytes(), sizes

    stream, sizes = shard.compress_sharded(data, oracle_compress, dist, 1, 131072)
    want, wsizes = oracle_compress(data, 1, 131072)

    # the decode direction: frames are sharded the same way (frame ranges per rank, ordered gather of the contents);
    # frame walking is the product's host code (zhip_find_frames), the per-rank decoder a test double (the oracle)
    def oracle_decompress(buf):
        buf = np.frombuffer(bytes(buf), dtype=np.uint8)
        out = np.empty(n + 64, dtype=np.uint8)
        r = lo.zo_decompress(_buf(out), len(out), _buf(buf), len(buf))
        assert r != ERR
        return out[:r].tobytes()
    content = shard.decompress_sharded(want, oracle_decompress, dist)
    if rank == 0:
        q.put((stream == want and content == data.tobytes(), bool(np.array_equal(sizes, wsizes)), len(stream)))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = q.get(timeout=300)
    for p in ps:
        p.join(timeout=300)
        assert p.exitcode == 0
    return res


def test_two_ranks_ordered_gather_equals_single_process():
    same, sizes_ok, length = _run(2, 131072 * 7 + 999)
    assert same and sizes_ok and length > 0


def test_two_ranks_empty_input_is_one_empty_frame():
    """an empty source is one (empty) unit: exactly one rank compresses it, the stream equals the single-process one"""
    same, sizes_ok, length = _run(2, 0)
    assert same and sizes_ok and length > 0


def test_unit_ranges_partition():
    from zstd_amd import shard
    for n_units in (0, 1, 2, 7, 8, 8192, 8193):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = shard.unit_range(n_units, world, r)
                assert 0 <= lo <= hi <= n_units
                got += list(range(lo, hi))
            assert got == list(range(n_units))


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher must start two ranks itself and print ONE line with n_gpus = 2
    (stub context: gloo, no GPU work — ZHIP_BENCH_STUB=1); a --gpus that disagrees with WORLD_SIZE must fail loudly"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["ZHIP_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["data"] == "stub"
    env["WORLD_SIZE"] = "1"; env["RANK"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "disagrees" in (r.stderr + r.stdout)


def test_eight_ranks_ordered_gather_equals_single_process():
    """the first 8-GPU lease must not fail on plumbing (round-5 verdict, item 7): eight ranks, unit ranges, ordered host gather — with fewer units than ranks in
    one case (ranks with an empty range)"""
    same, sizes_ok, length = _run(8, 131072 * 19 + 4321)
    assert same and sizes_ok and length > 0
    same, sizes_ok, length = _run(8, 131072 * 3 + 5)
    assert same and sizes_ok and length > 0


def test_bench_eight_ranks_the_way_the_driver_launches_them():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...` (the contract's launch line)
    in the stub context: ONE JSON line from rank 0, n_gpus = 8"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["ZHIP_BENCH_STUB"] = "1"
    port = 29500 + (os.getpid() % 2000) + 7
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["data"] == "stub"
