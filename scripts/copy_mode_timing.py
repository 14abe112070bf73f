#!/usr/bin/env python3
"""scripts/copy_mode_timing.py — GPU box, measurement helper: throughput of the dictionary COPY mode (sources above the reference's attach
cut-off, k_ext_init + k_parse_ext: one LANE per source) against the attach mode and against the reference on the host, same dictionary."""
import ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import zstd_amd
from zstd_amd import workloads as W
zd = np.fromfile(os.path.join(ROOT, "tests", "golden", "github_like_110k.zdict"), dtype=np.uint8)
flat, offs = W.github_like_records_native(600000, seed=5)
dev = torch.device("cuda", 0)
exe = os.path.join(ROOT, "oracle", "_ref", "zref_bench")
for rec_bytes, nrec in ((1200, 500000), (12000, 50000), (24000, 25000), (48000, 12000), (100000, 6000)):
    per = max(1, rec_bytes // 1200)
    nrec = min(nrec, 600000 // per)
    o = offs[np.arange(nrec + 1) * per]                       # records = `per` consecutive JSON records glued together
    n = int(o[-1])
    src = torch.empty(n + 64, dtype=torch.uint8, device=dev); src[:n].copy_(torch.from_numpy(flat[:n]))
    o64 = np.ascontiguousarray(o - o[0], dtype=np.uint64)
    cap = int(zstd_amd.lib().zhip_records_bound(o64.ctypes.data_as(C.c_void_p), nrec))
    dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    ctx = zstd_amd.Context(0, max_units=nrec, records_total_bytes=n)
    cd = zstd_amd.CDict(zd, level=3, device=0)
    best = 1e9
    for _ in range(3):
        tot = ctx.compress_records_device(cd, dst.data_ptr(), cap, src.data_ptr(), o64)
        t = ctx.timing(); best = min(best, t["total_ms"])
    line = {"record_bytes": int(n // nrec), "records": nrec, "GBps_device": round(n / best / 1e6, 2), "parse_ms": round(t["parse_ms"], 2), "entropy_ms": round(t["entropy_ms"], 2), "ratio": round(n / tot, 3)}
    if os.path.exists(exe):
        zd.tofile("/tmp/cm_d.bin"); flat[:n].tofile("/tmp/cm_r.bin"); o64.astype("<u8").tofile("/tmp/cm_o.bin")
        one = json.loads(subprocess.check_output([exe, "dict", "3", "/tmp/cm_d.bin", "/tmp/cm_r.bin", "/tmp/cm_o.bin", "2", "1"]))
        allc = json.loads(subprocess.check_output([exe, "dict", "3", "/tmp/cm_d.bin", "/tmp/cm_r.bin", "/tmp/cm_o.bin", "2", str(os.cpu_count())]))
        line.update(ref_1core_GBps=round(one["MBps"] / 1e3, 3), ref_all_GBps=round(allc["MBps"] / 1e3, 2), ref_ratio=one["ratio"])
    print(json.dumps(line), flush=True)
    ctx.close(); del src, dst
