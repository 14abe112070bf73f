#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X zstd block-compression core (contract: see the task statement).

Workload (BASELINE.json configs[1]): `datagen -g1073741824 -P50 -s<rank>` (1 GiB, programs/datagen.c stream mode),
level 1 (ZSTD_fast), 131072-byte independent units, one frame per unit, source and destination resident in HBM.
One *step* = one pass of the whole device pipeline (match finder -> entropy/frame assembly -> compaction) over the
1 GiB batch.  `value` = source MB (1e6 bytes, programs/benchzstd.h:32) per second, whole job over all ranks.

N > 1: one process per GPU — under torchrun, or started by bench.py itself when `--gpus N` is given without a launcher — every rank compresses its own 1 GiB shard of independent units on its own
GPU/stream; there is no data-path collective (units are independent) — torch.distributed only provides the
barriers and the max-over-ranks time.  scaling = "weak".

Besides the contract fields the JSON line carries `roofline` (dominant kernel = the match finder stage k_parse_fast_q / k_parse_fast_g: algorithmic
bytes = source read once + 8-byte sequence records + literals written once, divided by the stage's average
duration from HIP events recorded by the library on the stream it launches on — the two kernels run side by side on two streams and share one
ticket queue, so the STAGE is what the events bracket), `pipeline` (all kernels, (S + C) bytes),
`ratio`, `parity` (sha256 of the GPU stream == the real reference's stream at full size, and the first units against the C restatement),
`cpu_baseline` (the REAL reference timed on this box's host cores).

The line is printed as soon as the headline leg is done.  Every further leg (decode, pipelined, end_to_end, multi_block_frames, job_pool_frame,
silesia_shaped_level1, silesia64_level3 = BASELINE configs[2], records_zdict_level3 = configs[4], level5_row_prediction) then runs as a CHILD process
under its own deadline, and the augmented line is printed again after each: the LAST line of stdout is the complete one, and a leg that stalls or
dies leaves {"error": ...} under its key instead of taking the line with it (round 3 lost every number of the round to one stalled leg).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
UNIT = 131072


def cpu_baseline(sample, seconds=8.0, level=1):
    """time the reference (oracle/_ref/zref_bench, built from /root/reference) on the host: `zstd -b1 -B128K` semantics"""
    exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
    tmp = "/tmp/zhip_bench_sample.bin"
    if os.path.exists(exe):
        sample.tofile(tmp)
        try:
            one = json.loads(subprocess.check_output([exe, "file", str(level), str(UNIT), tmp, str(seconds), "1"], timeout=120))
            ncores = os.cpu_count() or 1
            # the all-core figure moved 2.7-9.1 GB/s between runs at level 5 (round-5 verdict): three separate processes (thread placement is decided at start), the best counts
            allr = [json.loads(subprocess.check_output([exe, "file", str(level), str(UNIT), tmp, str(max(1.0, seconds / 4)), str(ncores)], timeout=120)) for _ in range(3)]
            allc = max(allr, key=lambda r: r["MBps"])
            extra = {}
            if level >= 5:      # both matchers exist on the device; `value` is the reference default (row hash), the extra figure its hash-chain mode
                env = dict(os.environ, ZREF_NOROW="1")
                hc1 = json.loads(subprocess.check_output([exe, "file", str(level), str(UNIT), tmp, str(seconds / 2), "1"], timeout=120, env=env))
                hca = json.loads(subprocess.check_output([exe, "file", str(level), str(UNIT), tmp, str(seconds / 2), str(ncores)], timeout=120, env=env))
                extra = {"hash_chain_mode": {"value": hc1["MBps"], "cores": 1, "ratio": hc1["ratio"], "all_cores": hca["MBps"],
                                             "note": "ZSTD_c_useRowMatchFinder=disable (ZHIP_ROW_MATCHER=disable on the device); `value` above is the reference default (row matcher), which the device reproduces by default"}}
            return {**extra, "value": one["MBps"], "unit": "MB/s", "cores": 1, "kind": "reference", "ratio": one["ratio"],
                    "sample": f"first {len(sample) >> 20} MiB of the workload, level {level}, {UNIT} B units, best of {one['runs']} runs "
                              f"(oracle/_ref/zref_bench = ZSTD_compress2 per unit, programs/benchzstd.c semantics)",
                    "all_cores": {"value": allc["MBps"], "cores": ncores, "best_of_processes": [r["MBps"] for r in allr]}}
        finally:
            os.unlink(tmp)
    # reference build absent: time our C restatement instead (slower than the real thing; say so)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import load_oracle, _buf
    lo = load_oracle()
    cap = lo.zo_compress_bound(UNIT) * (len(sample) // UNIT + 1)
    dst = np.empty(cap, dtype=np.uint8)
    t0 = time.time()
    r = lo.zo_compress_chunks(level, UNIT, _buf(sample), len(sample), _buf(dst), cap, None, 0)
    dt = time.time() - t0
    return {"value": len(sample) / dt / 1e6, "unit": "MB/s", "cores": 1, "kind": "port", "ratio": len(sample) / r,
            "sample": f"first {len(sample) >> 20} MiB, oracle/zoracle.c single pass"}


def cpu_decode_baseline(sample, seconds=6.0, level=1):
    """the reference's DECODE speed on the host (`zstd -b#` second figure): oracle/_ref/zref_bench dfile"""
    exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
    if not os.path.exists(exe):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _libs import load_oracle, _buf
        lo = load_oracle()
        sample = sample[: 32 << 20]
        cap = lo.zo_compress_bound(UNIT) * (len(sample) // UNIT + 1)
        comp = np.empty(cap, dtype=np.uint8)
        r = lo.zo_compress_chunks(level, UNIT, _buf(sample), len(sample), _buf(comp), cap, None, 0)
        out = np.empty(len(sample), dtype=np.uint8)
        t0 = time.time()
        k = lo.zo_decompress(_buf(out), len(out), _buf(comp), r)
        dt = time.time() - t0
        assert k == len(sample)
        return {"value": len(sample) / dt / 1e6, "unit": "MB/s", "cores": 1, "kind": "port", "sample": f"first {len(sample) >> 20} MiB, oracle/zoracle_dec.c single pass"}
    tmp = "/tmp/zhip_bench_dsample.bin"
    sample.tofile(tmp)
    try:
        one = json.loads(subprocess.check_output([exe, "dfile", str(level), str(UNIT), tmp, str(seconds), "1"], timeout=120))
        ncores = os.cpu_count() or 1
        allc = json.loads(subprocess.check_output([exe, "dfile", str(level), str(UNIT), tmp, str(seconds / 2), str(ncores)], timeout=120))
        return {"value": one["MBps"], "unit": "MB/s", "cores": 1, "kind": "reference",
                "sample": f"first {len(sample) >> 20} MiB of the workload compressed at level {level} into {UNIT} B-unit frames, ZSTD_decompressDCtx per frame, best of "
                          f"{one['runs']} runs (oracle/_ref/zref_bench dfile, linked against the reference built WITH its x86-64 assembly Huffman loops)",
                "all_cores": {"value": allc["MBps"], "cores": ncores}}
    finally:
        os.unlink(tmp)


def decode_measure(torch, zstd_amd, local, src, n, dst, sizes, steps, warmup, barrier):
    """decode the frames in dst (sizes = per-unit compressed sizes) back into a fresh buffer; returns (seconds for `steps` passes,
    mean k_decode ms, parity flag)"""
    units = len(sizes)
    csz = sizes.astype(np.uint64)
    so = np.concatenate([[0], np.cumsum(csz)[:-1]]).astype(np.uint64)
    do = np.arange(units, dtype=np.uint64) * UNIT
    dcap = np.full(units, UNIT, dtype=np.uint64)
    dcap[-1] = n - (units - 1) * UNIT
    out = torch.empty(n + 64, dtype=torch.uint8, device=src.device)
    dctx = zstd_amd.DContext(local)

    def dstep():
        return dctx.decompress_frames_device(out.data_ptr(), do, dcap, dst.data_ptr(), so, csz)
    for _ in range(warmup):
        dstep()
    barrier()
    t0 = time.perf_counter()
    kms = 0.0
    for _ in range(steps):
        r, status, dsz = dstep()
        kms += dctx.timing()["decode_ms"]
    barrier()
    dt = time.perf_counter() - t0
    ok = bool(r == n and not status.any() and torch.equal(out[:n], src[:n]))
    dctx.close()
    return dt, kms / steps, ok


def parity_check(host, n, dev_out, total, sizes, level=1, tile=None):
    """byte parity of the GPU stream.  With oracle/_ref present: the WHOLE workload is compressed once by the real reference on all
    host cores (zref_bench cfile) and the SHA-256 of that stream is compared with the SHA-256 of the whole GPU stream — full size.
    Always: the first 64 units byte for byte against the C restatement, and the structure of every frame (magic at the offset the
    size table implies, sizes add up).  `host` may be shorter than n when the workload is tiled on the device: `tile` = (base corpus,
    shift) then lets the reference build the very same tiling in memory (zref_bench ctile) — still the whole stream, full size."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _libs import load_oracle, _buf, ERR
    lo = load_oracle()
    lo.zo_set_row_matcher.argtypes = [C.c_int]
    lo.zo_set_row_matcher(0 if os.environ.get("ZHIP_ROW_MATCHER") in ("disable", "0") else 1)     # the matcher the device context uses
    nsamp = min(64, len(host) // UNIT)
    sample = host[: nsamp * UNIT]
    cap = lo.zo_compress_bound(UNIT) * max(1, nsamp)
    dst = np.empty(cap, dtype=np.uint8)
    osz = np.zeros(max(1, nsamp), dtype=np.uint64)
    r = lo.zo_compress_chunks(level, UNIT, _buf(sample), len(sample), _buf(dst), cap, _buf(osz), nsamp) if nsamp else 0
    assert r != ERR
    gpu_prefix = dev_out[: int(r)].cpu().numpy()
    same = hashlib.sha256(gpu_prefix.tobytes()).hexdigest() == hashlib.sha256(dst[:r].tobytes()).hexdigest()
    same = same and np.array_equal(sizes[:nsamp].astype(np.uint64), osz[:nsamp])
    offs = np.concatenate([[0], np.cumsum(sizes.astype(np.int64))])
    idx = torch.from_numpy(offs[:-1]).to(dev_out.device)
    magic_ok = bool(((dev_out[idx] == 0x28) & (dev_out[idx + 1] == 0xB5) & (dev_out[idx + 2] == 0x2F) & (dev_out[idx + 3] == 0xFD)).all())
    res = {"bytes_identical_to_oracle_first_64_units": bool(same), "frames_well_formed": magic_ok and int(offs[-1]) == int(total)}
    exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
    if os.path.exists(exe):
        tiled = tile is not None and len(host) < n
        full = n if tiled else (min(len(host), n) // UNIT * UNIT if len(host) < n else min(len(host), n))
        tin, tout = f"/tmp/zhip_parity_in_{os.getpid()}.bin", f"/tmp/zhip_parity_out_{os.getpid()}.bin"
        try:
            # levels >= 5 (row-hash matcher): a new CCtx per unit — the salt of a reused CCtx depends on what it compressed before
            env = dict(os.environ, ZREF_FRESH_CCTX="1") if level >= 5 else dict(os.environ)
            env.pop("ZREF_NOROW", None)
            nthr = str(os.cpu_count() or 1)
            if tiled:
                base, shift = tile
                base.tofile(tin)
                info = json.loads(subprocess.check_output([exe, "ctile", str(level), str(UNIT), tin, "0", str(shift), str(n), tout, nthr], timeout=900, env=env))
            else:
                host[:full].tofile(tin)
                info = json.loads(subprocess.check_output([exe, "cfile", str(level), str(UNIT), tin, tout, nthr], timeout=600, env=env))
            h = hashlib.sha256()
            with open(tout, "rb") as f:
                for blk in iter(lambda: f.read(1 << 26), b""):
                    h.update(blk)
            nu = (full + UNIT - 1) // UNIT
            glen = int(offs[nu])
            gh = hashlib.sha256()
            for a0 in range(0, glen, 1 << 28):                  # the GPU stream, hashed in 256 MiB pieces
                gh.update(dev_out[a0: min(glen, a0 + (1 << 28))].cpu().numpy())
            g = gh.hexdigest()
            res["full_size"] = {"sha256_equals_reference_stream": bool(g == h.hexdigest() and glen == info["csize"]), "source_bytes": int(full),
                                "compressed_bytes": glen, "units": int(nu), "sha256": g,
                                "reference": f"oracle/_ref/zref_bench {'ctile (the same tiling built in memory)' if tiled else 'cfile'} = ZSTD_compress2 per unit on {info['threads']} host threads, {info['seconds']} s"}
            if tiled and "MBps" in info:
                res["full_size"]["reference_all_threads_MBps"] = info["MBps"]
        finally:
            for t in (tin, tout):
                if os.path.exists(t):
                    os.unlink(t)
    return res


def records_leg(args, torch, zstd_amd, dev, local, rank, world, dist, n_records, distinct, steps, warmup, want_cpu, want_decode):
    """BASELINE configs[4] stand-in: `n_records` ~1.2 KB JSON records (GitHub-user shaped), `distinct` of them different (generated natively,
    zstd_amd/workloads_src) and tiled, each its own frame, compressed with a dictionary attached (ZSTD_createCDict + refCDict + compress2
    per record).  Dictionary = the committed ZDICT-trained fixture (or, with --raw-dict, the first ~110 KB of records as raw content).
    One step = every record once.  Parity at FULL size: SHA-256 of the whole GPU stream against the real reference's stream of the
    distinct records (zref_bench cdict) repeated once per copy; and the first 256 frames byte for byte against the C restatement."""
    from zstd_amd import workloads as W
    level = args.level if args.level != 1 else 3                       # configs[4] is level 3; --level 1 is the bench default
    distinct = max(1, min(distinct, n_records))
    flat, offs = W.github_like_records_native(distinct, seed=rank)
    zpath = os.path.join(ROOT, "tests", "golden", "github_like_110k.zdict")
    if os.path.exists(zpath) and not args.raw_dict:
        dict_ = np.fromfile(zpath, dtype=np.uint8)                      # ZDICT_trainFromBuffer on 4 000 such records (tests/golden/make_dict.py)
        ddesc = f"ZDICT-trained dictionary of {len(dict_)} B (tests/golden/github_like_110k.zdict, made with the reference's ZDICT_trainFromBuffer)"
    else:
        ndict = int(np.searchsorted(offs, 110 * 1024))
        dict_ = flat[: int(offs[ndict])].copy()
        ddesc = f"raw-content dictionary of {len(dict_)} B (the first records)"
    base_n, L = len(offs) - 1, int(offs[-1])
    copies = max(1, n_records // base_n)
    n = L * copies
    all_offs = (np.arange(copies, dtype=np.uint64)[:, None] * np.uint64(L) + offs[None, :-1]).reshape(-1)
    all_offs = np.concatenate([all_offs, [np.uint64(n)]]).astype(np.uint64)
    nrec = base_n * copies
    bdev = torch.from_numpy(flat).to(dev)
    src = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    for c in range(copies):
        src[c * L:(c + 1) * L].copy_(bdev)
    del bdev
    cap = int(zstd_amd.lib().zhip_records_bound(all_offs.ctypes.data_as(C.c_void_p), nrec))
    dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    fsz = torch.zeros(nrec, dtype=torch.int32, device=dev)
    ctx = zstd_amd.Context(local, max_units=nrec, records_total_bytes=n)
    cd = zstd_amd.CDict(dict_, level=level, device=local)

    def step():
        return ctx.compress_records_device(cd, dst.data_ptr(), cap, src.data_ptr(), all_offs, fsz.data_ptr())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        total = step()
    barrier()
    t0 = time.perf_counter()
    kp = ke = kt = 0.0
    for _ in range(steps):
        total = step()
        tm = ctx.timing(); kp += tm["parse_ms"]; ke += tm["entropy_ms"]; kt += tm["total_ms"]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    out = None
    if rank == 0:
        K = steps
        sizes = fsz.cpu().numpy()
        # parity 1: the first 256 frames byte for byte against the oracle's CDict restatement
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _libs import load_oracle, _buf, ERR
        lo = load_oracle()
        lo.zo_cdict_create.restype = C.c_void_p; lo.zo_cdict_create.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        lo.zo_compress_unit_cdict.restype = C.c_size_t; lo.zo_compress_unit_cdict.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        ocd = lo.zo_cdict_create(_buf(dict_), len(dict_), level)
        nchk = min(256, base_n)
        gpu = dst[: int(sizes[:nchk].sum())].cpu().numpy().tobytes()
        want = b""
        for i in range(nchk):
            r = flat[int(offs[i]): int(offs[i + 1])]
            buf = np.zeros(len(r) + 700, dtype=np.uint8)
            k = lo.zo_compress_unit_cdict(_buf(buf), len(buf), _buf(r), len(r), ocd)
            assert k != ERR
            want += buf[:k].tobytes()
        parity = {"bytes_identical_to_oracle_first_256_records": gpu == want}
        # parity 2, full size: the whole GPU stream against the real reference's stream of the distinct records, once per copy
        exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
        pid = os.getpid()
        fd, fr, fo, fout = (f"/tmp/zb_{pid}_{x}.bin" for x in ("d", "r", "o", "out"))
        have_ref = os.path.exists(exe)
        if have_ref:
            dict_.tofile(fd); flat.tofile(fr); offs.astype("<u8").tofile(fo)
        try:
            if have_ref:
                nthr = os.cpu_count() or 1
                info = json.loads(subprocess.check_output([exe, "cdict", str(level), fd, fr, fo, fout, str(nthr)], timeout=600))
                ref_stream = np.fromfile(fout, dtype=np.uint8)
                h = hashlib.sha256()
                for _ in range(copies):
                    h.update(ref_stream)
                gh = hashlib.sha256()
                glen = int(total)
                for a0 in range(0, glen, 1 << 28):
                    gh.update(dst[a0: min(glen, a0 + (1 << 28))].cpu().numpy())
                parity["full_size"] = {"sha256_equals_reference_stream": bool(gh.hexdigest() == h.hexdigest() and glen == copies * int(info["csize"])),
                                       "records": int(nrec), "distinct_records": int(base_n), "source_bytes": int(n), "compressed_bytes": glen, "sha256": gh.hexdigest(),
                                       "reference": f"oracle/_ref/zref_bench cdict = ZSTD_createCDict + refCDict + ZSTD_compress2 of every distinct record on {info['threads']} host "
                                                    f"threads ({info['seconds']} s), the stream hashed once per copy"}
            parse_ms, ent_ms, tot_ms = kp / K, ke / K, kt / K
            algo = n + int(total)
            traffic, tsrc = traffic_lookup("records_zdict_level3", "k_parse_dict")
            out = {"metric": f"compress_MBps_level{level}_records_with_dictionary", "value": round(world * n / dt * K / 1e6, 1), "unit": "MB/s",
                   "n_gpus": world, "steps": K, "warmup": warmup, "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "u8/u32 integer", "data": "synthetic",
                   "config": {"workload": f"{nrec} JSON records (GitHub-user shaped, mean {L // base_n} B; {base_n} distinct = {L} B, tiled x{copies}), one frame per record, "
                                          f"level {level}, {ddesc} attached (ZSTD_createCDict + refCDict + compress2 semantics), src+dst in HBM",
                              "records_per_gpu": nrec, "parallelism": f"{world} x (one process per GPU, independent records, no collective)"},
                   "ratio": round(n / float(total), 4),
                   "roofline": {"bound": "hbm", "kernel": "k_parse_dict_q (+ k_parse_dict_g on the same ticket queue)", "achieved": round(algo / (parse_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(algo / (parse_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                                "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(parse_ms, 3)},
                   "pipeline": {"parse_ms": round(parse_ms, 3), "entropy_ms": round(ent_ms, 3), "device_total_ms": round(tot_ms, 3),
                                "host_ms_per_step": round(dt / K * 1e3 - tot_ms, 3)},
                   "parity": parity}
            # the CPU figures use a bounded sample: the first 100 000 distinct records (about 120 MB)
            ns = min(base_n, 100000)
            if have_ref and (want_cpu or want_decode):
                flat[: int(offs[ns])].tofile(fr); offs[: ns + 1].astype("<u8").tofile(fo)
            if want_cpu and have_ref:
                one = json.loads(subprocess.check_output([exe, "dict", str(level), fd, fr, fo, "5", "1"], timeout=120))
                nc = os.cpu_count() or 1
                allc = json.loads(subprocess.check_output([exe, "dict", str(level), fd, fr, fo, "3", str(nc)], timeout=120))
                out["cpu_baseline"] = {"value": one["MBps"], "unit": "MB/s", "cores": 1, "kind": "reference", "ratio": one["ratio"],
                                       "sample": f"the first {ns} distinct records, same dictionary, ZSTD_createCDict + refCDict + compress2 per record (oracle/_ref/zref_bench dict)",
                                       "all_cores": {"value": allc["MBps"], "cores": nc}}
            if want_decode and world == 1:                      # the way back: every record frame decoded with the dictionary (ZSTD_decompress_usingDDict per record)
                dd = zstd_amd.DDict(dict_, device=local)
                dctx = zstd_amd.DContext(local)
                csz = sizes.astype(np.uint64)
                so = np.concatenate([[0], np.cumsum(csz)[:-1]]).astype(np.uint64)
                rsz = np.diff(all_offs).astype(np.uint64)
                back = torch.empty(n + 64, dtype=torch.uint8, device=dev)
                best = 1e9
                for _ in range(2):
                    r, status, dsz = dctx.decompress_frames_device(back.data_ptr(), all_offs[:-1], rsz, dst.data_ptr(), so, csz, ddict=dd)
                    best = min(best, dctx.timing()["decode_ms"])
                okd = bool(r == n and not status.any() and torch.equal(back[:n], src[:n]))
                del back
                dcpu = None
                if want_cpu and have_ref:
                    try:
                        one = json.loads(subprocess.check_output([exe, "ddict", str(level), fd, fr, fo, "4", "1"], timeout=120))
                        nc = os.cpu_count() or 1
                        allc = json.loads(subprocess.check_output([exe, "ddict", str(level), fd, fr, fo, "3", str(nc)], timeout=120))
                        dcpu = {"value": one["MBps"], "unit": "MB/s", "cores": 1, "kind": "reference",
                                "sample": f"the first {ns} distinct record frames, ZSTD_createDDict + ZSTD_decompress_usingDDict per record (oracle/_ref/zref_bench ddict)",
                                "all_cores": {"value": allc["MBps"], "cores": nc}}
                    except Exception:
                        dcpu = None
                out["decode"] = {"metric": "decompress_MBps_records_with_dictionary", "value": round(n / best / 1e3, 1), "unit": "MB/s", "k_decode_ms": round(best, 3),
                                 "roofline": {"bound": "hbm", "kernel": "k_decode", "achieved": round(algo / (best * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": round(algo / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": None},
                                 "parity": {"decoded_equals_source_full_size": okd}}
                if dcpu:
                    out["decode"]["cpu_baseline"] = dcpu
        finally:
            for t in (fd, fr, fo, fout):
                if os.path.exists(t):
                    os.unlink(t)
    ctx.close()
    del src, dst, fsz
    return out


def traffic_lookup(leg, kernel):
    """HBM bytes per launch of `kernel` in bench leg `leg` from the committed PMC summary (profiles/latest_traffic.json) — measured
    by rocprofv3 --pmc passes of that very configuration, not in this run; (None, None) when the file has no entry"""
    tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
    try:
        tj = json.load(open(tpath))
        e = tj.get("legs", {}).get(leg, {})
        v = e.get(kernel + "_hbm_bytes_per_launch")
        if v is None:
            return None, None
        return v, "profiles/latest_traffic.json <- " + str(e.get("source", tj.get("source", "")))
    except Exception:
        return None, None


def records_main(args, torch, zstd_amd, dev, local, rank, world, dist):
    nrec = args.records if args.records else max(args.base_records, ((args.mib << 20) // 1201))
    out = records_leg(args, torch, zstd_amd, dev, local, rank, world, dist, nrec, args.base_records, args.steps, args.warmup,
                      want_cpu=not args.no_cpu_baseline, want_decode=True)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


def real_corpus(env_name):
    """the REAL corpus when the box has one: $ZHIP_SILESIA (the Silesia corpus as one file, e.g. silesia.tar — tests/regression/data.c:33-52 names its
    members) / $ZHIP_ENWIK9 give a file path; None when unset.  No box of this project ever had either: the hooks exist so that the first one that does
    yields the metric's real number (round-5 verdict, item 6 ii); tests/test_bench_corpus_hooks.py exercises them with a small file."""
    path = os.environ.get(env_name)
    if not path:
        return None
    if not os.path.isfile(path) or os.path.getsize(path) < 4096:
        raise SystemExit(f"bench.py: ${env_name}={path} is not a readable file of at least 4 KB")
    return np.fromfile(path, dtype=np.uint8)


def lorem_corpus(n, seed=0):
    """LOREM_genBuffer(buffer, n, seed) — what `zstd -b#` compresses when it is given no file (programs/benchzstd.c:1014, programs/lorem.h:20; SURVEY 8(d) names it as
    the text stand-in) — made by the reference's own generator in oracle/_ref/libzstd_ref.so.  Generation only: the bytes are input data, nothing of the reference runs
    in the timed path."""
    import ctypes as C
    path = os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so")
    if not os.path.exists(path):
        raise SystemExit("bench.py: the lorem workload needs oracle/_ref/libzstd_ref.so (make -C oracle ref, where /root/reference exists)")
    lr = C.CDLL(path)
    lr.LOREM_genBuffer.restype = None
    lr.LOREM_genBuffer.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
    a = np.zeros(n, dtype=np.uint8)
    lr.LOREM_genBuffer(a.ctypes.data_as(C.c_void_p), n, seed)
    return a


def make_workload(torch, zstd_amd, dev, workload, rank, world, mib, copies, total_bytes):
    """-> (host array for the CPU legs, device source tensor, n, description, scaling, tile = (base corpus, shift) when the host array is only the base).
    make_workload.data_kind says what the bytes are: "synthetic", or "real" when $ZHIP_SILESIA / $ZHIP_ENWIK9 supplied the corpus."""
    scaling = "weak"
    make_workload.data_kind = "synthetic"
    if workload == "datagen":
        n = mib << 20
        host = zstd_amd.datagen(n, 50, seed=rank, stream_mode=True)        # `datagen -g<n> -P50 -s<rank>`
        src = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        src[:n].copy_(torch.from_numpy(host))
        return host, src, n, f"datagen -g{n} -P50 -s<rank> (programs/datagen.c stream mode)", scaling, None
    from zstd_amd import workloads as W
    if workload == "silesia":                                               # configs[0]/[2]: Silesia is not on disk -> Silesia-shaped stand-in, unless $ZHIP_SILESIA names the corpus
        base = real_corpus("ZHIP_SILESIA")
        if base is not None:
            make_workload.data_kind = "real"
            wdesc = f"the Silesia corpus ($ZHIP_SILESIA = {os.environ['ZHIP_SILESIA']}, {len(base)} B) x{copies} copies"
        else:
            base = W.silesia_like(lambda m, P, seed: zstd_amd.datagen(m, P, seed=seed, stream_mode=False), seed=rank)
            wdesc = f"Silesia-shaped synthetic mix ({len(base)} B: text / structured / tables / runs / incompressible, zstd_amd/workloads.py) x{copies} copies"
        n = len(base) * copies
    else:                                                                   # configs[3]: enwik9 is not on disk -> Zipf word-salad text (or LOREM_genBuffer), unless $ZHIP_ENWIK9 names it
        real = real_corpus("ZHIP_ENWIK9") if workload == "text" else None
        if real is not None:
            base = real; make_workload.data_kind = "real"
        elif workload == "lorem":
            base = lorem_corpus(64 << 20, seed=rank)
        else:
            base = W.text_corpus(64 << 20, seed=rank)
        if total_bytes:
            n = total_bytes // world                                        # frame-per-shard: a fixed total cut into one shard per GPU
            scaling = "strong"
        else:
            n = mib << 20
        if real is not None:
            wdesc = f"enwik9 ($ZHIP_ENWIK9 = {os.environ['ZHIP_ENWIK9']}, {len(base)} B; {n} B per GPU, tiled if shorter)"
        elif workload == "lorem":
            wdesc = f"LOREM_genBuffer text (programs/lorem.h:20 — what `zstd -b#` compresses without a file; 64 MiB from the reference's generator, tiled to {n} B per GPU)"
        else:
            wdesc = f"Zipf word-salad text (64 MiB generated, tiled to {n} B per GPU, zstd_amd/workloads.py), enwik9 stand-in"
    bdev = torch.from_numpy(base).to(dev)
    src = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    pos, c, L = 0, 0, len(base)
    while pos < n:                                                          # copy c starts at offset c*9973: units of different copies differ
        s0 = (c * 9973) % L
        for a0, a1 in ((s0, L), (0, s0)):
            take = min(a1 - a0, n - pos)
            if take > 0:
                src[pos:pos + take].copy_(bdev[a0:a0 + take]); pos += take
        c += 1
    del bdev
    host = src[:n].cpu().numpy() if n <= (3 << 30) else base[: min(len(base), n)]     # the CPU legs see exactly what the device compresses
    return host, src, n, wdesc, scaling, ((base, 9973) if n > (3 << 30) else None)


def compress_leg(args, torch, zstd_amd, dev, local, rank, world, dist, workload, level, steps, warmup, copies, total_bytes,
                 want_decode, want_pipelined, want_cpu, cpu_seconds=8.0, leg=None):
    """one workload through the device pipeline: K timed steps bracketed by barrier + synchronize, max over ranks; rank 0 returns the
    JSON object (contract fields + roofline + pipeline + parity + cpu_baseline [+ decode, pipelined]), other ranks None"""
    host, src, n, wdesc, scaling, tile = make_workload(torch, zstd_amd, dev, workload, rank, world, args.mib, copies, total_bytes)
    units = (n + UNIT - 1) // UNIT
    cap = zstd_amd.compress_bound(n, UNIT)
    dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    usz = torch.zeros(units, dtype=torch.int32, device=dev)
    ctx = zstd_amd.Context(local, max_units=units)

    def step():
        return ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, level, UNIT, usz.data_ptr())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        total = step()
    barrier()
    t0 = time.perf_counter()
    kparse = kent = kgat = ktot = 0.0
    hc = {"chain_ms": 0.0, "search_ms": 0.0, "parse_ms": 0.0}
    for _ in range(steps):
        total = step()
        tm = ctx.timing()
        kparse += tm["parse_ms"]; kent += tm["entropy_ms"]; kgat += tm["gather_ms"]; ktot += tm["total_ms"]
        for k, v in ctx.hc_timing().items():
            hc[k] += v
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ct = torch.tensor([float(total)], dtype=torch.float64)
        dist.all_reduce(ct, op=dist.ReduceOp.SUM)
        total_all = float(ct.item())
    else:
        total_all = float(total)

    sizes_all = usz.cpu().numpy()
    ddt, dkms, dok = (None, None, None)
    if want_decode:
        dsteps = steps if args.mode == "decode" else max(2, min(steps, 4))
        ddt, dkms, dok = decode_measure(torch, zstd_amd, local, src, n, dst, sizes_all, dsteps, warmup if args.mode == "decode" else 1, barrier)
        if dist is not None:
            tt = torch.tensor([ddt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ddt = float(tt.item())
    out = None
    if rank == 0:
        st = ctx.stats()
        sizes = sizes_all
        K = steps
        ms_step = dt / K * 1e3
        parse_ms, ent_ms, gat_ms, tot_ms = kparse / K, kent / K, kgat / K, ktot / K
        # algorithmic bytes: the unit is read once (S) and its result written once.  For the match
        # finder alone the result is the 8-byte sequence records; for the pipeline it is the compressed stream (C).
        parse_bytes = n + 8 * st["sequences"] + st["literals"]
        achieved = parse_bytes / (parse_ms * 1e-3) / 1e9
        cp = zstd_amd.get_cparams(level, UNIT)
        traffic, tsrc = None, None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if os.path.exists(tpath) and workload == "datagen" and level == 1 and n == (1 << 30):
            try:                                                # PMC passes of this very configuration, committed with their summaries
                tj = json.load(open(tpath))
                traffic = tj.get("k_parse_fast_hbm_bytes_per_launch")
                tsrc = "profiles/latest_traffic.json <- " + str(tj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration (not measured in this run)"))
            except Exception:
                traffic, tsrc = None, None
        elif leg is not None:
            traffic, tsrc = traffic_lookup(leg, {1: "k_parse_fast", 2: "k_parse_dfast"}.get(zstd_amd.get_cparams(level, UNIT)[6], "k_parse_lazy"))
        sname = {1: "ZSTD_fast", 2: "ZSTD_dfast", 3: "ZSTD_greedy (hash chain)", 4: "ZSTD_lazy (hash chain)", 5: "ZSTD_lazy2 (hash chain)"}[cp[6]]
        cpdesc = f"{sname} wlog{cp[0]} clog{cp[1]} hlog{cp[2]} slog{cp[3]} mml{cp[4]}"
        # ZSTD_fast runs as ONE stage of two kernels on one ticket queue (LDS-table wavefronts + global-table wavefronts beside them, zhip_lib.hip
        # launch_parse); the stage's duration — cost estimate and sort of the dispatch order included — is what the events bracket
        kname = {1: "k_parse_fast_q (+ k_parse_fast_g on the same queue; k_order_cost/k_order_sort inside the stage)",
                 2: "k_parse_dfast"}.get(cp[6], "k_hc_chain+k_hc_search+k_parse_lazy")
        out = {
            "metric": f"compress_MBps_level{level}_{'datagenP50' if workload == 'datagen' else workload}_128KB_units", "value": round(world * n / dt * K / 1e6, 1), "unit": "MB/s",
            "n_gpus": world, "steps": K, "warmup": warmup, "ms_per_step": round(ms_step, 3),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "u8/u32 integer", "data": getattr(make_workload, "data_kind", "synthetic"),
            "config": {"workload": f"{wdesc}, level {level} ({cpdesc}), "
                                   f"{UNIT} B independent units = one frame each, src+dst resident in HBM", "units_per_gpu": units,
                       "parallelism": f"{world} x (one process per GPU, independent units, no collective)"},
            "ratio": round(world * n / total_all, 4),
            # algorithmic bytes = S + C (source read once, compressed stream written once) over the dominant kernel's duration;
            # the same kernel priced at its own stage's bytes (S + 8 * nbSeq + L) is kept beside it as frac_stage_bytes
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round((n + int(total)) / (parse_ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round((n + int(total)) / (parse_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": tsrc,
                         "algorithmic_bytes_per_launch": n + int(total), "avg_launch_ms": round(parse_ms, 3),
                         "frac_stage_bytes": round(achieved / HBM_PEAK_GBS, 5), "stage_bytes_per_launch": parse_bytes},
            "pipeline": {"parse_ms": round(parse_ms, 3), "entropy_ms": round(ent_ms, 3), "gather_ms": round(gat_ms, 3),
                         "device_total_ms": round(tot_ms, 3), "algorithmic_bytes": n + int(total),
                         "achieved_GBps": round((n + int(total)) / (tot_ms * 1e-3) / 1e9, 2),
                         "frac_of_hbm_peak": round((n + int(total)) / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if cp[6] >= 3:      # the match-finder stage is three kernels; the roofline figures above are for their sum
            out["roofline"]["kernels_ms"] = {k: round(v / K, 3) for k, v in hc.items()}
        out["parity"] = parity_check(host, n, dst, total, sizes, level, tile) if not args.no_parity else {"skipped": "--no-parity (profiling run)"}
        if want_pipelined and os.environ.get("ZHIP_PIPELINE_CHUNKS") is None and units <= 16384:
            # optional stream pipelining of the same workload (a second context: the knob is read at creation)
            os.environ["ZHIP_PIPELINE_CHUNKS"] = "4"
            ctx2 = zstd_amd.Context(local, max_units=units)
            del os.environ["ZHIP_PIPELINE_CHUNKS"]
            for _ in range(2):
                t2 = ctx2.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, level, UNIT, usz.data_ptr())
            torch.cuda.synchronize(); q0 = time.perf_counter()
            for _ in range(3):
                t2 = ctx2.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, level, UNIT, usz.data_ptr())
            torch.cuda.synchronize(); q1 = time.perf_counter()
            out["pipelined"] = {"chunks": 4, "value": round(n / ((q1 - q0) / 3) / 1e6, 1), "unit": "MB/s", "steps": 3,
                                "same_bytes": bool(int(t2) == int(total))}
            ctx2.close()
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(host[: 256 << 20] if len(host) >= (256 << 20) else host, seconds=cpu_seconds, level=level)
        if ddt is not None:
            dK = steps if args.mode == "decode" else max(2, min(steps, 4))
            dbytes = n + int(total)                             # algorithmic bytes of k_decode: the frames in, their content out
            dtraffic = None
            if traffic is not None:
                try:
                    dtraffic = json.load(open(tpath)).get("k_decode_hbm_bytes_per_launch")
                except Exception:
                    dtraffic = None
            dec = {"metric": f"decompress_MBps_level{level}_{'datagenP50' if workload == 'datagen' else workload}_128KB_units",
                   "value": round(world * n / ddt * dK / 1e6, 1), "unit": "MB/s", "steps": dK, "ms_per_step": round(ddt / dK * 1e3, 3),
                   "roofline": {"bound": "hbm", "kernel": "k_decode", "achieved": round(dbytes / (dkms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": round(dbytes / (dkms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": dtraffic, "traffic_source": tsrc if dtraffic is not None else None,
                                "algorithmic_bytes_per_launch": dbytes, "avg_launch_ms": round(dkms, 3)},
                   "parity": {"decoded_equals_source_full_size": dok}}
            if want_cpu and world == 1:
                dec["cpu_baseline"] = cpu_decode_baseline(host[: 256 << 20] if len(host) >= (256 << 20) else host, level=level)
            if args.mode == "decode":                           # the decoder is the headline: same contract fields, the compressor's line rides along
                comp_line = {k: out[k] for k in ("metric", "value", "unit", "ms_per_step", "ratio", "roofline", "parity") if k in out}
                out.update({"metric": dec["metric"], "value": dec["value"], "ms_per_step": dec["ms_per_step"], "steps": dK, "roofline": dec["roofline"],
                            "parity": dec["parity"], "compress": comp_line})
                if "cpu_baseline" in dec:
                    out["cpu_baseline"] = dec["cpu_baseline"]
                out.pop("pipeline", None); out.pop("pipelined", None)
            else:
                out["decode"] = dec
    ctx.close()
    del dst, usz
    return out, (host, src, n, int(total))


def end_to_end_leg(torch, zstd_amd, local, host, total_expected, level):
    """SURVEY.md §8(d) second timing: host buffers in, host bytes out (H2D + kernels + D2H + gather) through zhip_compress_multi /
    zhip_compress — PCIe-inclusive, never `value`.  The multi-lane path is measured FIRST, before this process has created any other context."""
    units = (len(host) + UNIT - 1) // UNIT
    m = zstd_amd.MultiContext([local])
    dst = np.empty(zstd_amd.compress_bound(len(host)), dtype=np.uint8)
    bestm, k = 1e9, 0
    for _ in range(4):
        t0 = time.perf_counter()
        k = m.compress_into(dst, host, level=level)
        bestm = min(bestm, time.perf_counter() - t0)
    stages = m.last_stages()
    multi_stream = dst[:k].tobytes()
    # the same call on four times the source (the buffer tiled: units are independent, so the output is the 1x output four times): what is left of
    # the ramp — first chunk in, last chunk's kernels and copy-out, about 8 ms — weighs a quarter as much
    big = None
    if len(host) % UNIT == 0 and len(host) <= (1 << 30):
        host4 = np.tile(host, 4)
        dst4 = np.empty(zstd_amd.compress_bound(len(host4)), dtype=np.uint8)
        b4, k4 = 1e9, 0
        for _ in range(3):
            t0 = time.perf_counter()
            k4 = m.compress_into(dst4, host4, level=level)
            b4 = min(b4, time.perf_counter() - t0)
        same4 = bool(k4 == 4 * k and all(dst4[i * k:(i + 1) * k].tobytes() == multi_stream for i in range(4)))
        big = {"value": round(len(host4) / b4 / 1e6, 1), "unit": "MB/s", "source_bytes": int(len(host4)), "best_of": 3, "same_bytes_as_device_path": same4,
               "stages_of_last_call": m.last_stages()}
        del host4, dst4
    m.close()
    # the drop-in itself: ZSTD_compress2 through libzstd_hipshim.so on the same source (what programs/benchzstd.c:344 calls).  From 256 MiB on the shim routes the call through
    # the lanes above (round 6; the plain single-stream call before); must be within 10 % of the multi-lane figure and the same bytes
    shim = None
    try:
        if len(host) % (1 << 20) == 0:
            o = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "shim_compress2_timing.py"), str(len(host) >> 20), str(level), str(local)], timeout=120, stderr=subprocess.DEVNULL)
            sj = json.loads([l for l in o.decode().splitlines() if l.startswith("{")][-1])
            shim = {"value": sj["value"], "unit": "MB/s", "best_of": sj["best_of"],
                    "same_bytes_as_multi_lane_call": bool(sj["bytes"] == k and sj["sha256"] == hashlib.sha256(multi_stream).hexdigest()),
                    "path": "a process of its own (scripts/shim_compress2_timing.py): ZSTD_createCCtx + ZSTD_CCtx_setParameter(ZSTD_c_compressionLevel) + ZSTD_compress2 of libzstd_hipshim.so "
                            "(include/zstd_hip_dropin.h) on the whole source; from 256 MiB on the shim runs the call on the lanes of zhip_compress_multi"}
    except Exception as e:                                       # noqa: BLE001
        shim = {"error": str(e)}
    ctx = zstd_amd.Context(local, max_units=units)
    best = 1e9
    got = None
    for _ in range(2):
        t0 = time.perf_counter()
        got = ctx.compress(host, level=level)
        best = min(best, time.perf_counter() - t0)
    ctx.close()
    same = bool((total_expected is None or k == int(total_expected)) and multi_stream == got)       # `got` = the single context's stream of the same source
    if big is not None:
        big["same_bytes_as_device_path"] = bool(big["same_bytes_as_device_path"] and same)
    return {"value": round(len(host) / bestm / 1e6, 1), "unit": "MB/s", "best_of": 4, "source_bytes": int(len(host)),
            "same_bytes_as_device_path": same,
            "stages_of_last_call": stages,
            "four_times_the_source": big,
            "dropin_ZSTD_compress2": shim,
            "path": "zhip_compress_multi on this one device: two lanes (kernel stream + copy stream, feeder / device / gatherer threads, two pinned slots each way; staging copies split over 4 host threads), 128 MB chunks with quarter "
                    "chunks at both ends: memcpy -> H2D (under the previous chunk's kernels) -> kernels -> D2H -> ordered host gather into the caller's buffer; stage seconds are summed "
                    "over chunks and lanes (they overlap); PCIe- and host-memcpy-inclusive, never `value`; the leg's process holds the library only (no torch tensors, no other context before the measurement)",
            "synchronous_single_stream": {"value": round(len(host) / best / 1e6, 1), "unit": "MB/s", "path": "zhip_compress: pageable source, blocking H2D / kernels / D2H on one stream"}}


def frames_leg(zstd_amd, local, host, level):
    """inputs of 1 MiB, ONE multi-block frame each (the reference's own output shape), a batch of up to 1024
    (two workgroups per CU: 512 run at once) through zhip_compress_frames.  The rate is over the frame kernel's duration (HIP events; the call itself goes through host
    buffers); parity = SHA-256 of all frames against ZSTD_compress2 of each 1 MiB input by the real reference, full size."""
    fsz, nf = 1 << 20, min(1024, len(host) >> 20)
    if nf == 0:
        return None
    bufs = [host[i * fsz:(i + 1) * fsz] for i in range(nf)]
    ctx = zstd_amd.Context(local, max_units=nf)
    best, outs = 1e9, None
    for _ in range(3):
        outs = ctx.compress_frames(bufs, level)
        best = min(best, ctx.timing()["entropy_ms"])
    ctx.close()
    algo = nf * fsz + sum(len(o) for o in outs)
    res = {"value": round(nf * fsz / best / 1e3, 1), "unit": "MB/s", "frames": nf, "frame_bytes": fsz, "level": level,
           "kernel_ms": round(best, 3), "ratio": round(nf * fsz / sum(len(o) for o in outs), 4),
           "roofline": {"bound": "hbm", "kernel": "k_frame_hbm" if nf > 512 else "k_frame_fast", "achieved": round(algo / (best * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(algo / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic_lookup("multi_block_frames", "k_frame_hbm")[0] if (nf, fsz) == (1024, 1 << 20) else None,
                        "traffic_source": traffic_lookup("multi_block_frames", "k_frame_hbm")[1] if (nf, fsz) == (1024, 1 << 20) else None,
                        "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(best, 3)},
           "note": "one workgroup per frame (blocks of a frame are a serial chain): k_frame_fast (table in LDS, two per CU) up to 2 x CUs frames, k_frame_hbm (tables in HBM, four per CU) beyond — round 6; fidelity mode, never `value`"}
    exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
    if os.path.exists(exe):
        tin, tout = f"/tmp/zhip_frames_in_{os.getpid()}.bin", f"/tmp/zhip_frames_out_{os.getpid()}.bin"
        try:
            host[: nf * fsz].tofile(tin)
            info = json.loads(subprocess.check_output([exe, "cfile", str(level), str(fsz), tin, tout, str(os.cpu_count() or 1)], timeout=600))
            want = hashlib.sha256(open(tout, "rb").read()).hexdigest()
            got = hashlib.sha256(b"".join(outs)).hexdigest()
            res["parity"] = {"sha256_equals_reference_frames": bool(got == want and info["csize"] == sum(len(o) for o in outs)),
                             "reference": f"oracle/_ref/zref_bench cfile = ZSTD_compress2 of every 1 MiB input (one multi-block frame each) on {info['threads']} host threads"}
        finally:
            for t in (tin, tout):
                if os.path.exists(t):
                    os.unlink(t)
    return res


def job_pool_leg(zstd_amd, local, host, level):
    """the WHOLE workload as ONE frame the way ZSTD_compress2 emits it with ZSTD_c_nbWorkers >= 1 (jobs of the
    default job size with the overlap as prefix; zhip_compress_frames_mt) — a workgroup per job, so one frame fills the GPU.  Rate
    over the frame kernel's duration; parity = SHA-256 against the real reference's job pool on all host threads, which is also the
    CPU figure beside it (`zstd -T0` on one big input)."""
    n = min(len(host), (1 << 31) - (1 << 20))
    if n <= (512 << 10):
        return None
    a = host[:n]
    ctx = zstd_amd.Context(local, max_units=max(64, n // (512 << 10) + 1))
    best, out = 1e9, None
    for _ in range(3):
        out = ctx.compress_frames([a], level, workers=1)[0]
        best = min(best, ctx.timing()["entropy_ms"])
    jobs = int(ctx.stats()["units"])
    # the same call from pageable host memory, wall clock: blocking H2D + kernels + D2H (what the drop-in's ZSTD_c_nbWorkers mode does)
    import ctypes as C
    L = zstd_amd.lib()
    L.zhip_compress_frames_mt.restype = C.c_size_t
    L.zhip_compress_frames_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    offs = np.array([0, n], dtype=np.uint64)
    d1 = np.empty(zstd_amd.compress_bound(n) + 64, dtype=np.uint8)
    wall = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = L.zhip_compress_frames_mt(ctx._h, d1.ctypes.data_as(C.c_void_p), d1.nbytes, a.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), 1, level, None, 0, 0, None)
        wall = min(wall, time.perf_counter() - t0)
    same = bool(not L.zhip_isError(r) and d1[:r].tobytes() == out)
    ctx.close()
    # the way back: that ONE frame through the decoder — block-parallel (zhip_decode_big.h), where one workgroup walked it at 0.26 GB/s
    dec = None
    try:
        d = zstd_amd.DContext(local)
        back, dms = None, 1e9
        for _ in range(2):
            back = d.decompress(out, capacity=n)
            dms = min(dms, d.timing()["decode_ms"])
        dec = {"value": round(n / dms / 1e3, 1), "unit": "MB/s", "decode_ms": round(dms, 3), "equals_source": bool(back == a.tobytes()), **d.last_bigframe(),
               "path": "zhip_decompress of the frame above: k_bf_walk / prep / deps / entropy (a workgroup per block) / scan / build / jump rounds / copy; device time between the first and the last launch, host round trips of the jump loop included"}
        d.close()
    except Exception as e:                                       # noqa: BLE001
        dec = {"error": str(e)}
    res = {"value": round(n / best / 1e3, 1), "unit": "MB/s", "frame_bytes": int(n), "jobs": jobs, "level": level, "kernel_ms": round(best, 3),
           "roofline": {"bound": "hbm", "kernel": "k_frame_fast (job table)", "achieved": round((n + len(out)) / (best * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round((n + len(out)) / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic_lookup("job_pool_frame", "k_frame_fast")[0] if n == (1 << 30) else None,
                        "traffic_source": traffic_lookup("job_pool_frame", "k_frame_fast")[1] if n == (1 << 30) else None,
                        "algorithmic_bytes_per_launch": int(n + len(out)), "avg_launch_ms": round(best, 3)},
           "end_to_end": {"value": round(n / wall / 1e6, 1), "unit": "MB/s", "same_bytes": same,
                          "path": "zhip_compress_frames_mt from pageable host memory: blocking H2D, kernels, D2H on one context; PCIe-inclusive, never `value`"},
           "ratio": round(n / len(out), 4), "decode": dec,
           "note": "k_frame_fast with a job table: one frame, jobs = independent workgroups, two per CU (48 KB 24-bit LDS table each); never `value`"}
    exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
    if os.path.exists(exe):
        tin, tout = f"/tmp/zhip_mt_in_{os.getpid()}.bin", f"/tmp/zhip_mt_out_{os.getpid()}.bin"
        try:
            a.tofile(tin)
            thr = os.cpu_count() or 1
            info = json.loads(subprocess.check_output([exe, "mtfile", str(level), str(min(thr, 256)), tin, tout, "0"], timeout=600))
            want = hashlib.sha256(open(tout, "rb").read()).hexdigest()
            res["parity"] = {"sha256_equals_reference_frame": bool(hashlib.sha256(out).hexdigest() == want and info["csize"] == len(out)),
                             "reference": f"oracle/_ref/zref_bench mtfile = ZSTD_compress2 of the whole input, ZSTD_c_nbWorkers = {info['workers']}"}
            res["cpu_reference"] = {"value": info["MBps"], "unit": "MB/s", "workers": info["workers"], "seconds": info["seconds"]}
        except (subprocess.CalledProcessError, subprocess.TimeoutExpired) as e:
            res["parity"] = {"error": str(e)}
        finally:
            for t in (tin, tout):
                if os.path.exists(t):
                    os.unlink(t)
    return res


def prediction_leg(zstd_amd, local, host):
    """The row matcher's two-pass prediction, measured where it matters: level 5 (greedy, the reference's default row-hash matcher) on the
    headline's long-match data, the same call with the prediction off (the default) and on.  Device time of the whole call; the two outputs must be the same bytes
    and the first units are checked against the oracle.  Never `value`."""
    n = len(host) // UNIT * UNIT
    if n < 16 * UNIT:
        return None
    a = host[:n]
    res = {"level": 5, "source_bytes": int(n), "units": int(n // UNIT),
           "note": "zhip_compress at level 5 (ZSTD_greedy, row-hash matcher), 128 KB units; `off` (the default) = one parse, every search behind a left-out position redone live from the row matcher's live rows (DESIGN.md 4.2b); `on` = zhip_set_prediction(units=1): tried, predicted, parsed again; device ms of the call (parse + entropy + gather), best of 2; `first_256MiB_off` = the same call on the first 2 048 units only (half of the 4 096 resident wavefronts: the size this leg was quoted on before); never `value`"}
    try:
        ctx = zstd_amd.Context(local, max_units=n // UNIT + 1)
        ctx.set_row_matcher(0)
        if n > (256 << 20):
            ctx.set_prediction(units=0)
            best = 1e9
            for _ in range(2):
                ctx.compress(a[: 256 << 20], level=5)
                best = min(best, ctx.timing()["total_ms"])
            res["first_256MiB_off"] = {"value": round((256 << 20) / best / 1e3, 1), "unit": "MB/s", "device_ms": round(best, 3)}
        outs = {}
        for mode in (0, 1):
            ctx.set_prediction(units=mode)
            best, hc, out = 1e9, None, None
            for _ in range(2):
                out = ctx.compress(a, level=5)
                t = ctx.timing()
                if t["total_ms"] < best:
                    best, hc = t["total_ms"], ctx.hc_timing()
            outs[mode] = out
            res["on" if mode else "off"] = {"value": round(n / best / 1e3, 1), "unit": "MB/s", "device_ms": round(best, 3), "match_finder_ms": {k: round(v, 3) for k, v in hc.items()}}
        ctx.close()
        res["same_bytes"] = bool(outs[0] == outs[1])
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from _libs import load_oracle, _buf, ERR
        lo = load_oracle()
        lo.zo_set_row_matcher.argtypes = [C.c_int]
        lo.zo_set_row_matcher(1)
        try:
            nsamp = 8
            cap = lo.zo_compress_bound(UNIT) * nsamp
            dst = np.empty(cap, dtype=np.uint8)
            r = lo.zo_compress_chunks(5, UNIT, _buf(a[: nsamp * UNIT]), nsamp * UNIT, _buf(dst), cap, None, 0)
            res["bytes_identical_to_oracle_first_8_units"] = bool(r != ERR and outs[1][: int(r)] == dst[: int(r)].tobytes())
        finally:
            lo.zo_set_row_matcher(0)
    except Exception as e:                                       # noqa: BLE001
        res["error"] = str(e)
    return res


def plugin_leg(zstd_amd, local, host, level):
    """Boundary B1 at the headline's size: zhip_prepare_sequences parses every 128 KB block of the workload on the device in ONE batch, then the REAL
    reference (oracle/_ref/libzstd_ref.so) compresses the same buffer with zhip_sequence_producer registered (ZSTD_registerSequenceProducer,
    lib/zstd.h:2866; contrib/externalSequenceProducer/main.c:38-49 shape) — its own entropy stage on N host threads, one CCtx and one shard of whole
    blocks each (a CCtx with a producer cannot use nbWorkers, lib/compress/zstd_compress.c:7180).  Reported: MB/s of prepare + compress, the device
    part's roofline at S + 16 * nbSeq, round trip through the reference's decoder, and sequence-level equality of every block with
    ZSTD_generateSequences of the reference on that block (full size).  Never `value`."""
    zpath = os.path.join(ROOT, "oracle", "_ref", "libzstd_ref.so")
    if not os.path.exists(zpath):
        return {"skipped": "oracle/_ref/libzstd_ref.so absent (built from /root/reference by oracle/Makefile)"}
    import threading
    Z = C.CDLL(zpath)
    Z.ZSTD_createCCtx.restype = C.c_void_p
    Z.ZSTD_freeCCtx.argtypes = [C.c_void_p]
    Z.ZSTD_CCtx_setParameter.restype = C.c_size_t; Z.ZSTD_CCtx_setParameter.argtypes = [C.c_void_p, C.c_int, C.c_int]
    Z.ZSTD_registerSequenceProducer.restype = None; Z.ZSTD_registerSequenceProducer.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    Z.ZSTD_compress2.restype = C.c_size_t; Z.ZSTD_compress2.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    Z.ZSTD_decompress.restype = C.c_size_t; Z.ZSTD_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    Z.ZSTD_isError.restype = C.c_uint; Z.ZSTD_isError.argtypes = [C.c_size_t]
    Z.ZSTD_compressBound.restype = C.c_size_t; Z.ZSTD_compressBound.argtypes = [C.c_size_t]
    BLK = 65536            # ZSTD_c_maxBlockSize below 128 KB: the reference then asks for exactly these blocks (at 128 KB it splits some at 92 KB, zstd_compress.c:4494-4518)
    n = len(host) // BLK * BLK
    a = np.ascontiguousarray(host[:n])
    units = n // BLK
    nthr = max(1, min(64, os.cpu_count() or 1, units))
    per = (units + nthr - 1) // nthr * BLK                              # whole blocks per shard: every callback's block is one prepared block
    shards = [(o, min(per, n - o)) for o in range(0, n, per)]
    L = zstd_amd.lib()
    ctx = zstd_amd.Context(local, max_units=units)
    fn = C.cast(L.zhip_sequence_producer, C.c_void_p)
    base = a.ctypes.data
    dsts = [np.empty(int(Z.ZSTD_compressBound(ln)), dtype=np.uint8) for _, ln in shards]
    sizes = [0] * len(shards)

    def work(i):
        o, ln = shards[i]
        cctx = Z.ZSTD_createCCtx()
        Z.ZSTD_CCtx_setParameter(cctx, 100, level)                      # ZSTD_c_compressionLevel
        Z.ZSTD_CCtx_setParameter(cctx, 1015, BLK)                       # ZSTD_c_maxBlockSize: the blocks the producer is asked for are the prepared ones
        Z.ZSTD_CCtx_setParameter(cctx, 1009, 1)                         # ZSTD_c_validateSequences
        Z.ZSTD_registerSequenceProducer(cctx, ctx._h, fn)
        sizes[i] = Z.ZSTD_compress2(cctx, dsts[i].ctypes.data, dsts[i].nbytes, base + o, ln)
        Z.ZSTD_freeCCtx(cctx)

    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        r = L.zhip_prepare_sequences(ctx._h, C.c_void_p(base), n, BLK, level)
        t1 = time.perf_counter()
        if L.zhip_isError(r):
            ctx.close()
            return {"error": "zhip_prepare_sequences failed"}
        dev_ms = ctx.timing()["parse_ms"]
        th = [threading.Thread(target=work, args=(i,)) for i in range(len(shards))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        t2 = time.perf_counter()
        cur = {"prepare_s": t1 - t0, "compress_s": t2 - t1, "device_parse_ms": dev_ms}
        if best is None or cur["prepare_s"] + cur["compress_s"] < best["prepare_s"] + best["compress_s"]:
            best = cur
    if any(Z.ZSTD_isError(x) for x in sizes):
        ctx.close()
        return {"error": "ZSTD_compress2 with the producer registered failed"}
    st = ctx.stats()
    csize = int(sum(sizes))
    # round trip through the reference's decoder, every shard
    ok_rt = True
    back = np.empty(per, dtype=np.uint8)
    for (o, ln), d, k in zip(shards, dsts, sizes):
        got = Z.ZSTD_decompress(back.ctypes.data, ln, d.ctypes.data, k)
        ok_rt = ok_rt and got == ln and bool(np.array_equal(back[:ln], a[o:o + ln]))
    # sequence-level equality, full size: the device's (litLength, matchLength, offset) of every block against ZSTD_generateSequences of the
    # reference on that block alone (a dedicated CCtx per call: oracle/ref_shim.c zref_sequences), compared as one SHA-256 over all blocks
    seq_eq = None
    shim = os.path.join(ROOT, "oracle", "_ref", "libzref_shim.so")
    if os.path.exists(shim):
        try:
            R = C.CDLL(shim)
            R.zref_sequences.restype = C.c_size_t
            R.zref_sequences.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            hg, hr = hashlib.sha256(), hashlib.sha256()
            cap = BLK // 3 + 16
            ref_blocks = [None] * units

            def refwork(lo_, hi_):
                buf = np.zeros((cap, 4), dtype=np.uint32)
                for u in range(lo_, hi_):
                    k = R.zref_sequences(level, base + u * BLK, BLK, buf.ctypes.data, cap)
                    ref_blocks[u] = buf[:k, :3].copy()
            tw = max(1, min(64, os.cpu_count() or 1))
            th = [threading.Thread(target=refwork, args=(units * t // tw, units * (t + 1) // tw)) for t in range(tw)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            for u in range(units):
                g = ctx.get_sequences(u, cap=cap)
                hg.update(np.ascontiguousarray(g[:, :3]).tobytes()); hr.update(np.ascontiguousarray(ref_blocks[u]).tobytes())
            seq_eq = hg.hexdigest() == hr.hexdigest()
        except Exception as e:                                       # noqa: BLE001
            seq_eq = "not checked: " + str(e)
    ctx.close()
    tot_s = best["prepare_s"] + best["compress_s"]
    algo = n + 16 * int(st["sequences"])
    return {"value": round(n / tot_s / 1e6, 1), "unit": "MB/s", "level": level, "source_bytes": int(n), "blocks": int(units), "block_bytes": BLK, "host_threads": len(shards),
            "ratio": round(n / csize, 4), "prepare_MBps": round(n / best["prepare_s"] / 1e6, 1), "reference_entropy_stage_MBps": round(n / best["compress_s"] / 1e6, 1),
            "prepare_ms": round(best["prepare_s"] * 1e3, 2), "compress_ms": round(best["compress_s"] * 1e3, 2),
            "roofline": {"bound": "hbm", "kernel": "match finder stage of zhip_prepare_sequences (k_parse_fast_q/_g)", "achieved": round(algo / (best["device_parse_ms"] * 1e-3) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / (best["device_parse_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                         "traffic": traffic_lookup("plugin_B1", "k_parse_fast")[0] if n == (1 << 30) else None, "traffic_source": traffic_lookup("plugin_B1", "k_parse_fast")[1] if n == (1 << 30) else None,
                         "algorithmic_bytes_per_launch": int(algo), "avg_launch_ms": round(best["device_parse_ms"], 3)},
            "parity": {"roundtrip_through_reference_decoder": bool(ok_rt), "sequences_equal_ZSTD_generateSequences_full_size": seq_eq},
            "path": "zhip_prepare_sequences (pageable H2D, match finder, one packed D2H of the sequences, host fingerprints) + ZSTD_compress2 of the real reference with "
                    "zhip_sequence_producer registered, one CCtx per shard of whole blocks on that many host threads; PCIe-inclusive, never `value`"}


def strong_text_leg(args, torch, zstd_amd, dev, local, rank, world, dist):
    """BASELINE configs[3] as north_star words it: a FIXED total (10^9 B of text, --total-bytes) cut into one shard of whole 128 KB units per rank,
    every rank compresses its shard on its own GPU and copies its frames straight to its offset of ONE host buffer (a shared-memory segment: the
    host-side gather is the D2H itself, the offsets come from an all_gather of the shard sizes) — no data-path collective.  One step = compress +
    size exchange + gather, max over ranks.  Parity per rank: SHA-256 of its stream against the real reference's stream of its shard; rank 0 also
    reports the SHA-256 of the gathered stream (tests/test_gpu_multi.py compares it with a single-process run).  Returns the object on rank 0."""
    from multiprocessing import shared_memory
    from zstd_amd import workloads as W
    total = int(args.total_bytes) if args.total_bytes else 1_000_000_000
    units_total = (total + UNIT - 1) // UNIT
    per = (units_total + world - 1) // world * UNIT
    lo, hi = min(total, rank * per), min(total, (rank + 1) * per)
    n = hi - lo
    host = W.tile_range(W.text_corpus(64 << 20, seed=0), lo, hi)            # the same buffer on every rank; each makes only its own shard of it
    src = torch.empty(max(n, 1) + 64, dtype=torch.uint8, device=dev)
    if n:
        src[:n].copy_(torch.from_numpy(host))
    units = max(1, (n + UNIT - 1) // UNIT)
    cap = zstd_amd.compress_bound(max(n, 1), UNIT)
    dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    ctx = zstd_amd.Context(local, max_units=units)
    name = f"zhip_gather_{os.environ.get('MASTER_PORT', '0')}_{os.getppid() if world > 1 else os.getpid()}"
    bound_total = zstd_amd.compress_bound(total, UNIT) + 64 * world
    shm = None
    if rank == 0:
        try:
            old = shared_memory.SharedMemory(name=name); old.close(); old.unlink()
        except FileNotFoundError:
            pass
        shm = shared_memory.SharedMemory(name=name, create=True, size=bound_total)
    if dist is not None:
        dist.barrier()
    if rank != 0:
        shm = shared_memory.SharedMemory(name=name)
    gathered = torch.frombuffer(shm.buf, dtype=torch.uint8)

    def step():
        c = int(ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, 1, UNIT)) if n else 0
        sz = torch.tensor([c], dtype=torch.int64)
        if dist is not None:
            allsz = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(allsz, sz)
            sizes = [int(t.item()) for t in allsz]
        else:
            sizes = [c]
        off = sum(sizes[:rank])
        if c:
            gathered[off:off + c].copy_(dst[:c])                          # D2H into the shared host buffer = this rank's part of the gather
        return sizes

    K, Wm = max(2, min(args.steps, 10)), max(1, min(args.warmup, 2))
    for _ in range(Wm):
        sizes = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        sizes = step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # per-rank parity against the real reference (its threads shared out over the ranks)
    mine = gathered[sum(sizes[:rank]): sum(sizes[:rank + 1])].numpy()
    ok = None
    exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
    if os.path.exists(exe) and n and not args.no_parity:
        tin, tout = f"/tmp/zhip_strong_in_{os.getpid()}.bin", f"/tmp/zhip_strong_out_{os.getpid()}.bin"
        try:
            host.tofile(tin)
            info = json.loads(subprocess.check_output([exe, "cfile", "1", str(UNIT), tin, tout, str(max(1, (os.cpu_count() or 1) // world))], timeout=600))
            want = hashlib.sha256(open(tout, "rb").read()).hexdigest()
            ok = bool(hashlib.sha256(mine.tobytes()).hexdigest() == want and info["csize"] == len(mine))
        finally:
            for t in (tin, tout):
                if os.path.exists(t):
                    os.unlink(t)
    oks = [ok]
    if dist is not None:
        oks = [None] * world
        dist.all_gather_object(oks, ok)
    res = None
    if rank == 0:
        ctot = sum(sizes)
        res = {"metric": "compress_MBps_level1_text_frame_per_shard", "value": round(total / (dt / K) / 1e6, 1), "unit": "MB/s", "n_gpus": world, "steps": K,
               "ms_per_step": round(dt / K * 1e3, 3), "scaling": "strong", "ratio": round(total / max(1, ctot), 4),
               "config": {"workload": f"Zipf word-salad text (zstd_amd/workloads.py text_corpus, 64 MiB generated, tiled to {total} B; enwik9 stand-in), level 1, "
                                      f"{world} shard(s) of whole {UNIT} B units = independent frames, one shard per GPU", "shard_bytes": int(per)},
               "gather": "every rank's D2H lands at its offset of ONE shared host buffer (offsets from an all_gather of the shard sizes); timed inside the step",
               "parity": {"per_rank_sha256_equals_reference": oks, "full_size": {"sha256_equals_reference_stream": (all(bool(x) for x in oks) if all(x is not None for x in oks) else None)},
                          "gathered_stream_sha256": hashlib.sha256(gathered[:ctot].numpy().tobytes()).hexdigest(), "gathered_bytes": int(ctot)}}
    if dist is not None:
        dist.barrier()
    del gathered, mine
    ctx.close()
    try:
        shm.close()
        if rank == 0:
            shm.unlink()
    except Exception:                                             # noqa: BLE001
        pass
    return res


def stub_main(args, rank, world):
    """ZHIP_BENCH_STUB=1: no GPU, no compression — exercises only the launch / barrier / max-over-ranks / one-line contract of the
    N-rank path with the gloo backend (tests/test_dist_gloo.py); the line says data = "stub" and must never be read as a measurement"""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
    n = 1 << 20
    for _ in range(args.warmup):
        pass
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    acc = 0
    for _ in range(args.steps):
        acc += int(np.arange(n, dtype=np.uint8).sum())
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": round(world * n * args.steps / dt / 1e6, 1), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "u8", "data": "stub", "config": {"workload": "stub (no GPU work)"}}))
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mib", type=int, default=1024, help="source MiB per GPU (default = the 1 GiB of configs[1])")
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--workload", choices=["datagen", "silesia", "text", "lorem", "records"], default="datagen",
                    help="datagen = BASELINE configs[1] (default); silesia / text / records = synthetic stand-ins for configs[2] / [3] / [4]")
    ap.add_argument("--raw-dict", action="store_true", help="records: use the first ~110 KB of records as a raw-content dictionary instead of the trained fixture")
    ap.add_argument("--base-records", type=int, default=1000000, help="records: DISTINCT ~1.2 KB records generated before tiling (1 M = 1.2 GB, beyond the 256 MB Infinity Cache)")
    ap.add_argument("--records", type=int, default=0, help="records: total records per GPU (default: --mib worth; BASELINE configs[4] names 10 M)")
    ap.add_argument("--copies", type=int, default=1, help="silesia: number of copies of the 212 MB corpus (configs[2] uses 64)")
    ap.add_argument("--total-bytes", type=int, default=0, help="text: fixed total cut into one shard per GPU (configs[3]: 1000000000)")
    ap.add_argument("--mode", choices=["compress", "decode"], default="compress",
                    help="decode: the headline value is the DECODER's throughput on the frames the compressor just made (same workload, same units)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="profiling runs only: skip the parity legs (the printed line then says so and is not a measurement to quote)")
    ap.add_argument("--no-pipelined-extra", action="store_true", help="skip the extra 4-chunk pipelined measurement (profiling runs: keeps the per-kernel averages clean)")
    ap.add_argument("--leg", default="", help="internal: run ONE extra leg of the default line (see LEGS) and print its JSON — the default line starts these as child processes with deadlines")
    ap.add_argument("--budget", type=float, default=600.0, help="default line: wall-clock seconds after which no further extra leg is started (each leg also has its own deadline)")
    ap.add_argument("--no-extra-legs", action="store_true", help="default line only: skip the Silesia-shaped level-1 leg and the end-to-end (PCIe-inclusive) figure")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) and let rank 0 print the line
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world} (launch N ranks for --gpus N, or omit the launcher)")
    if os.environ.get("ZHIP_BENCH_STUB") == "1":
        return stub_main(args, rank, world)

    if args.leg == "end_to_end" and world == 1:
        # the host-buffer leg runs as a host application would: its process holds the library and nothing else — no torch tensors, no other context's
        # arenas (in a process that had run the device path before it, the same call measured 29 instead of 36 GB/s: profiles/r05_e2e_stages.log)
        import zstd_amd
        host = zstd_amd.datagen(args.mib << 20, 50, seed=rank, stream_mode=True)      # the headline's workload (make_workload)
        print(json.dumps(end_to_end_leg(None, zstd_amd, local, host, None, args.level)), flush=True)
        return

    import torch
    import zstd_amd

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (barriers, the max-over-ranks time): gloo over 127.0.0.1 — the data path has no collective and needs no RCCL
        dist.init_process_group("gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: zstd_amd has no CPU path")
    local_dev = local % torch.cuda.device_count()                # more ranks than visible GPUs: they share them (tests/test_gpu_multi.py runs two ranks on one device)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)

    if args.leg:                                                 # child of the default line: one extra leg in its own process (so that it can be given a deadline)
        print(json.dumps(run_leg(args, torch, zstd_amd, dev, local_dev)), flush=True)
        return
    if args.workload == "records":
        return records_main(args, torch, zstd_amd, dev, local_dev, rank, world, dist)
    default_line = (args.workload == "datagen" and args.level == 1 and args.mode == "compress" and world == 1 and not args.no_extra_legs)
    t_start = time.time()
    out, keep = compress_leg(args, torch, zstd_amd, dev, local_dev, rank, world, dist, args.workload, args.level, args.steps, args.warmup,
                             args.copies, args.total_bytes, want_decode=(args.mode == "decode" or (world == 1 and not default_line)),
                             want_pipelined=(world == 1 and not args.no_pipelined_extra and not default_line), want_cpu=not args.no_cpu_baseline)
    del keep
    if world > 1 and args.workload == "datagen" and not args.no_extra_legs:
        # BASELINE configs[3] beside the weak-scaling line: 10^9 B of text cut into one shard of whole units per rank, host gather timed
        ts = strong_text_leg(args, torch, zstd_amd, dev, local_dev, rank, world, dist)
        if rank == 0:
            out["text_strong_scaling"] = ts
    if rank == 0:
        out["digest"] = make_digest(out)
        print(json.dumps(out), flush=True)                      # the headline (metric, value, roofline, cpu_baseline, parity) is out before anything else runs
    if default_line and rank == 0:
        # Everything below is extra: each leg is a child process with its own deadline, and the line is printed again (augmented) after every
        # leg — the LAST line is the complete one, and a leg that stalls or dies leaves {"error": ...} instead of taking the line with it.
        torch.cuda.empty_cache()
        for name, deadline in LEGS:
            left = args.budget - (time.time() - t_start)
            if left < 20:
                out.pop("digest", None)
                out[name] = {"skipped": f"the run's time budget (--budget {args.budget} s) was used up"}
                out["digest"] = make_digest(out)
                continue
            out.pop("digest", None)
            out[name] = child_leg(name, min(deadline, left), args)
            out["digest"] = make_digest(out)                    # always the last key: the tail of the line alone says what every leg did
            print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# the default line's extra legs: (key in the JSON line, deadline in seconds)
LEGS = [("decode", 90), ("pipelined", 60), ("end_to_end", 90), ("plugin_B1", 120), ("multi_block_frames", 90), ("job_pool_frame", 120), ("silesia_shaped_level1", 120),
        ("text_level1", 150), ("lorem_level1", 120), ("silesia64_level3", 200), ("records_zdict_level3", 200), ("level5_units", 180), ("silesia_shaped_level5", 150)]
# short names of the legs in the line's closing `digest`
DIGEST_NAMES = {"decode": "dec_L1", "pipelined": "pipe4", "end_to_end": "e2e_host", "plugin_B1": "plugin_B1", "multi_block_frames": "frames_1MiB",
                "job_pool_frame": "job_pool_1GiB", "silesia_shaped_level1": "silesia4_L1", "text_level1": "text1e9_L1", "lorem_level1": "lorem1GiB_L1", "silesia64_level3": "silesia64_L3",
                "records_zdict_level3": "records10M_L3", "level5_units": "datagen_L5", "silesia_shaped_level5": "silesia4_L5"}


def make_digest(out):
    """every leg in one compact object, appended as the LAST key of the line: [MB/s, frac = (S + C) / dominant-kernel time / 8 TB/s, full-size parity
    against the real reference (sha256 of the stream; the decoder: decoded == source; plugin: sequences == ZSTD_generateSequences), the reference on
    ALL host cores in MB/s where the leg timed it]"""
    def ent(o):
        if not isinstance(o, dict):
            return None
        if "value" not in o:
            return [None, None, o.get("error") or o.get("skipped") or "no value"]
        par = o.get("parity", {}) if isinstance(o.get("parity"), dict) else {}
        ok = None
        for path in (("full_size", "sha256_equals_reference_stream"), ("sha256_equals_reference_frames",), ("sha256_equals_reference_frame",),
                     ("decoded_equals_source_full_size",), ("sequences_equal_ZSTD_generateSequences_full_size",)):
            cur = par
            for k in path:
                cur = cur.get(k) if isinstance(cur, dict) else None
            if cur is not None:
                ok = cur
                break
        if ok is None and "same_bytes_as_device_path" in o:
            ok = o["same_bytes_as_device_path"]
        if ok is None and "same_bytes" in o:
            ok = o["same_bytes"]
        host = None
        cb = o.get("cpu_baseline") or o.get("cpu_reference")
        if isinstance(cb, dict):
            host = (cb.get("all_cores") or {}).get("value") if "all_cores" in cb else cb.get("value")
        r = o.get("roofline") if isinstance(o.get("roofline"), dict) else {}
        return [o["value"], r.get("frac"), ok, host]
    d = {"fields": "[MB/s, frac_S_plus_C_of_8TB/s, parity_vs_reference_full_size, reference_all_host_cores_MB/s]", "datagen_L1": ent(out)}
    for name, _ in LEGS:
        if name in out:
            d[DIGEST_NAMES.get(name, name)] = ent(out[name])
            sh = out[name].get("dropin_ZSTD_compress2") if isinstance(out[name], dict) else None
            if name == "end_to_end" and isinstance(sh, dict) and "value" in sh:      # the drop-in's ZSTD_compress2 on the same source, a process of its own
                d["shim_compress2"] = [sh["value"], None, sh.get("same_bytes_as_multi_lane_call"), None]
    if isinstance(out.get("text_strong_scaling"), dict):
        d["text1e9_strong"] = ent(out["text_strong_scaling"])
    return d
LEG_KEYS = ("metric", "value", "unit", "steps", "ms_per_step", "ratio", "config", "roofline", "pipeline", "parity", "cpu_baseline")


def child_leg(name, deadline, args):
    cmd = [sys.executable, os.path.abspath(__file__), "--leg", name, "--steps", str(args.steps), "--warmup", str(args.warmup), "--mib", str(args.mib)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_parity:
        cmd.append("--no-parity")
    t0 = time.time()
    try:
        cp = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=deadline, text=True)
        line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
        res = json.loads(line[-1]) if line else {"error": f"no result (rc {cp.returncode})", "stderr_tail": cp.stderr[-400:]}
    except subprocess.TimeoutExpired:
        res = {"error": f"did not finish within {int(deadline)} s (child killed)"}
    except Exception as e:                                       # noqa: BLE001
        res = {"error": str(e)}
    if isinstance(res, dict):
        res["leg_wall_s"] = round(time.time() - t0, 1)
    return res


def run_leg(args, torch, zstd_amd, dev, local):
    """one extra leg, in the child: builds its own workload (same generator, same seed as the headline) and returns the leg's JSON object"""
    name = args.leg
    nocpu = args.no_cpu_baseline
    if name == "level5_row_prediction":
        host = zstd_amd.datagen(int(os.environ.get("ZHIP_L5_LEG_MIB", "1024")) << 20, 50, 0)
        return prediction_leg(zstd_amd, local, np.frombuffer(host, dtype=np.uint8) if not isinstance(host, np.ndarray) else host)
    if name == "silesia_shaped_level1":                          # the metric's own data shape: Silesia-shaped mix x4 at level 1
        sil, _ = compress_leg(args, torch, zstd_amd, dev, local, 0, 1, None, "silesia", 1, max(3, min(args.steps, 20)), 2, 4, 0,
                              want_decode=False, want_pipelined=False, want_cpu=not nocpu, cpu_seconds=6.0, leg="silesia4_level1")
        return {k: sil[k] for k in LEG_KEYS if k in sil}
    if name == "text_level1":                                    # BASELINE configs[3] on one GPU: 10^9 B of text as ONE shard, level 1
        tx, _ = compress_leg(args, torch, zstd_amd, dev, local, 0, 1, None, "text", 1, max(3, min(args.steps, 10)), 2, 1, 1000000000,
                             want_decode=False, want_pipelined=False, want_cpu=not nocpu, cpu_seconds=6.0, leg="text_level1")
        return {k: tx[k] for k in LEG_KEYS if k in tx}
    if name == "lorem_level1":                                   # SURVEY 8(d)'s text stand-in by name: LOREM_genBuffer, 1 GiB, level 1 (beside the word salad of text_level1)
        a6 = argparse.Namespace(**vars(args)); a6.mib = 1024
        lm, _ = compress_leg(a6, torch, zstd_amd, dev, local, 0, 1, None, "lorem", 1, max(3, min(args.steps, 10)), 2, 1, 0,
                             want_decode=False, want_pipelined=False, want_cpu=not nocpu, cpu_seconds=6.0, leg="lorem_level1")
        return {k: lm[k] for k in LEG_KEYS if k in lm}
    if name == "level5_units":                                   # the lazy family's sample row: level 5 (ZSTD_greedy, the reference's default row-hash matcher) on the headline's data
        host5 = int(os.environ.get("ZHIP_L5_LEG_MIB", "1024"))
        a5 = argparse.Namespace(**vars(args)); a5.mib = host5
        l5, _ = compress_leg(a5, torch, zstd_amd, dev, local, 0, 1, None, "datagen", 5, 3, 1, 1, 0,
                             want_decode=False, want_pipelined=False, want_cpu=not nocpu, cpu_seconds=6.0, leg="datagen_level5")
        return {k: l5[k] for k in LEG_KEYS if k in l5}
    if name == "silesia_shaped_level5":                          # the metric's data shape at the lazy family's level (round 6: its member of short runs took 4.4 s per unit until the 384-position rule stopped flagging a gap twice)
        s5, _ = compress_leg(args, torch, zstd_amd, dev, local, 0, 1, None, "silesia", 5, 3, 1, 4, 0,
                             want_decode=False, want_pipelined=False, want_cpu=not nocpu, cpu_seconds=6.0, leg="silesia4_level5")
        return {k: s5[k] for k in LEG_KEYS if k in s5}
    if name == "silesia64_level3":                               # BASELINE configs[2] at its stated size: the mix x64 (about 13 GiB), level 3 (ZSTD_dfast)
        sil3, _ = compress_leg(args, torch, zstd_amd, dev, local, 0, 1, None, "silesia", 3, 3, 1, 64, 0,
                               want_decode=False, want_pipelined=False, want_cpu=not nocpu, cpu_seconds=4.0, leg="silesia64_level3")
        return {k: sil3[k] for k in LEG_KEYS if k in sil3}
    if name == "records_zdict_level3":                           # BASELINE configs[4] at its stated size: 10 M records (1 M distinct), trained dictionary, level 3
        rec = records_leg(args, torch, zstd_amd, dev, local, 0, 1, None, 10_000_000, 1_000_000, 3, 1, want_cpu=not nocpu, want_decode=False)
        return {k: rec[k] for k in LEG_KEYS if k in rec}
    # the legs on the headline's own workload: a short pass of the device path first (its frames / its total are what they work on)
    args.no_parity = True
    short, (host, src, n, total) = compress_leg(args, torch, zstd_amd, dev, local, 0, 1, None, "datagen", args.level, 3, 1, 1, 0,
                                                 want_decode=(name == "decode"), want_pipelined=(name == "pipelined"), want_cpu=False)
    if name == "decode":
        dec = short["decode"]
        if not nocpu:
            dec["cpu_baseline"] = cpu_decode_baseline(host[: 256 << 20] if len(host) >= (256 << 20) else host, level=args.level)
        return dec
    if name == "pipelined":
        return short.get("pipelined", {"error": "not measured"})
    del src
    torch.cuda.empty_cache()
    if name == "end_to_end":
        return end_to_end_leg(torch, zstd_amd, local, host, total, args.level)
    if name == "plugin_B1":
        return plugin_leg(zstd_amd, local, host, args.level)
    if name == "multi_block_frames":
        return frames_leg(zstd_amd, local, host, args.level) or {"skipped": "workload below 1 MiB"}
    if name == "job_pool_frame":
        return job_pool_leg(zstd_amd, local, host, args.level) or {"skipped": "workload below 512 KB"}
    return {"error": "unknown leg " + name}


if __name__ == "__main__":
    main()
