#!/usr/bin/env python3
"""scripts/prof_decode.py — measurement helper (not product): per-phase s_memtime breakdown of k_decode on the bench workload
with the -DZHIP_PROF build (scripts/prof_phases.py build).  WORKLOAD=datagen|text, MIB, LEVEL."""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "zstd_amd", "libzstd_hip_prof.so")


def main():
    import torch
    import zstd_amd
    if os.environ.get("PROFLIB"):
        zstd_amd.LIB_PATH = os.path.join(ROOT, "zstd_amd", os.environ["PROFLIB"])
    elif os.path.exists(LIB) and not os.environ.get("NOPROF"):
        zstd_amd.LIB_PATH = LIB
    L = zstd_amd.lib()
    mib = int(os.environ.get("MIB", "256")); level = int(os.environ.get("LEVEL", "1"))
    n = mib << 20
    if os.environ.get("WORKLOAD", "datagen") == "text":
        from zstd_amd import workloads as W
        base = W.text_corpus(64 << 20, seed=0)
        host = np.concatenate([base] * (n // len(base) + 1))[:n]
    else:
        host = zstd_amd.datagen(n, 50, seed=0, stream_mode=True)
    src = torch.from_numpy(host).cuda()
    units = n // 131072
    ctx = zstd_amd.Context(0, max_units=units)
    cap = zstd_amd.compress_bound(n)
    comp = torch.empty(cap, dtype=torch.uint8, device="cuda")
    sizes = torch.empty(units, dtype=torch.int32, device="cuda")
    total = ctx.compress_device(comp.data_ptr(), cap, src.data_ptr(), n, level=level, sizes_ptr=sizes.data_ptr())
    csz = sizes.cpu().numpy().astype(np.uint64)
    so = np.concatenate([[0], np.cumsum(csz)[:-1]]).astype(np.uint64)
    do = np.arange(units, dtype=np.uint64) * 131072
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dctx = zstd_amd.DContext(0)
    prof = (C.c_ulonglong * 32)()
    have = hasattr(L, "zhip_prof_read")
    best = 1e9
    for it in range(4):
        if have:
            L.zhip_prof_read(prof, 1)
        r, status, dsz = dctx.decompress_frames_device(out.data_ptr(), do, np.full(units, 131072, np.uint64), comp.data_ptr(), so, csz)
        best = min(best, dctx.timing()["decode_ms"])
    assert r == n and torch.equal(out, src)
    res = {"mib": mib, "level": level, "ratio": n / total, "decode_ms": best, "GBps": n / best / 1e6}
    if have:
        L.zhip_prof_read(prof, 0)
        v = list(prof)
        w0 = ["setup", "literals", "wait(seq setup+chunk0)", "exec chunk", "wait(decode)", "last lits+barrier", "finish"]
        w1 = ["setup", "seq tables", "decode chunk0", "wait(literals)", "decode chunk", "wait(exec)", "last", "finish"]
        res["wave0_ticks_per_unit"] = {w0[i]: round(v[i] / units) for i in range(7)}
        res["wave1_ticks_per_unit"] = {w1[i]: round(v[16 + i] / units) for i in range(8)}
        names = ["ring refill+loop", "pass1 serial chain", "pass2 values", "repeat offsets", "scan+validate+store"]
        res["seq_chunk_ticks_per_unit"] = {names[i]: round(v[24 + i] / units) for i in range(5)}
        res["exec_rounds_per_batch"] = v[29] / max(1, v[30])
        res["seqs_per_unit"] = v[8] / units; res["lits_per_unit"] = v[9] / units
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
