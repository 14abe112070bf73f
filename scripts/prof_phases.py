#!/usr/bin/env python3
"""scripts/prof_phases.py — measurement helper (not product): builds zstd_amd/libzstd_hip_prof.so (-DZHIP_PROF) and prints
the per-phase s_memtime breakdown of k_parse_fast / k_entropy on the bench workload.  Build part runs anywhere
(`python scripts/prof_phases.py build`), the measurement needs a GPU."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "zstd_amd", "libzstd_hip_prof.so")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DZHIP_PROF",
           "-Wno-unused-result", os.path.join(ROOT, "zstd_amd", "csrc", "zhip_unity.hip"), "-o", LIB]      # one translation unit: the phase counters are one __device__ array
    subprocess.check_call(cmd)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
        return
    import torch
    import zstd_amd
    zstd_amd.LIB_PATH = LIB
    L = zstd_amd.lib()
    mib = int(os.environ.get("MIB", "1024"))
    level = int(os.environ.get("LEVEL", "1"))
    n = mib << 20
    if os.environ.get("WORKLOAD", "datagen") == "silesia":
        from zstd_amd import workloads as W
        host = W.tile(W.silesia_like(lambda size, P, seed: zstd_amd.datagen(size, P, seed=seed, stream_mode=False), seed=0), n)
    elif os.environ.get("WORKLOAD", "datagen") == "text":
        from zstd_amd import workloads as W
        import numpy as np
        base = W.text_corpus(64 << 20, seed=0)
        host = np.concatenate([base] * (n // len(base) + 1))[:n]
    else:
        host = zstd_amd.datagen(n, 50, seed=0, stream_mode=True)
    dev = torch.device("cuda", 0)
    src = torch.empty(n + 64, dtype=torch.uint8, device=dev)
    src[:n].copy_(torch.from_numpy(host))
    cap = zstd_amd.compress_bound(n, 131072)
    dst = torch.empty(cap + 64, dtype=torch.uint8, device=dev)
    ctx = zstd_amd.Context(0, max_units=n // 131072)
    out = (C.c_ulonglong * 32)()
    for it in range(3):
        ctx.compress_device(dst.data_ptr(), cap, src.data_ptr(), n, level, 131072)
        L.zhip_prof_read(out, 1)
        if it < 2 and level < 3: L.zhip_wph_read((C.c_ulonglong * 64)(), 1)
    v = list(out)
    units = n // 131072
    tm = ctx.timing()
    names_p = ["init", "window front (loads, hash, table + candidate gather, groups)", "window event search", "window match (E load, extension)",
               "window emit + inserts + repcode loop", "window end (table, sequences, literals)", "schedule-shaped batch", "its event (extension, post-match)", "tail",
               "window long match (post-match by loads)"]
    names_d = ["init (table zeroing)", "window: source + repcode bytes", "window: table gather (2 x 64 entries)", "window: scratch + candidate bytes (tag hits)",
               "window: cut W + hit masks", "window: event loop (E loads, extension, emits)", "window: end (table writes, sequences, literals)",
               "batch scheme until its event", "its match (df_extend) + emit", "post-match round(s) (inserts, immediate repcode, next bytes)"]
    names_e = ["gather+hist", "huf table build", "huf sizing+hdr", "huf pack", "seq hist+tables", "fse state chains",
               "seq bit pack", "headers"]
    if level >= 5:
        names_l = ["init", "batch front (records + repcode bytes, event detect)", "event pick + repcode extension", "search: LIVE (gap rule + list search)",
                   "search: from the record", "lazy steps + catch-up + prefetch", "emit (literals, sequence)", "immediate-repcode loop"]
        tot = sum(v[:8]) or 1
        print(json.dumps({"timing_ms": tm, "hc_ms": ctx.hc_timing(), "units": units,
                          "lazy_ticks_per_unit": {names_l[i]: [round(v[i] / units), round(100.0 * v[i] / tot, 1)] for i in range(8)},
                          "batches_per_unit": v[10] / units, "sequences_per_unit": v[11] / units, "live_searches_per_unit": v[12] / units, "failed_searches_per_unit": v[13] / units}, indent=1))
        return
    if level in (3, 4):
        tot = sum(v[:10]) or 1
        print(json.dumps({"timing_ms": tm, "units": units,
                          "dfast_ticks_per_unit": {names_d[i]: [round(v[i] / units), round(100.0 * v[i] / tot, 1)] for i in range(10)},
                          "event_loop_E_load_wait_ticks_per_unit (part of the event loop, not in the shares above)": round(v[15] / units),
                          "windows_per_unit": v[10] / units, "window_events_per_unit": v[11] / units, "batch_iterations_per_unit": v[12] / units,
                          "window_post_exits_per_unit": v[13] / units, "windows_cut_short_per_unit": v[14] / units}, indent=1))
        return
    w = (C.c_ulonglong * 64)()
    L.zhip_wph_read(w, 1)
    w = list(w)
    wn = {0: "F_SRC", 1: "F_TAB", 2: "F_DUP", 3: "F_GRP", 4: "F_MASK", 5: "SEARCH", 6: "M_ELOAD", 7: "M_RUNS", 8: "M_EMIT", 9: "LEAVE", 10: "E_PRE", 11: "E_TAB", 12: "E_OUT",
          13: "B_SCAN", 14: "B_MATCH", 15: "B_POST", 16: "TAIL", 17: "INIT", 18: "LOOP", 19: "CARRY", 20: "IMM", 21: "GRP_IT", 22: "GRP_NF", 23: "LATE", 24: "LEAVE_FAR"}
    wtot = sum(w[:32]) or 1
    res = {"timing_ms": tm, "units": units, "unit_bytes": n // units,
           # ZSTD_fast window phases (zhip_parse.h WPH_*): id -> [ticks per unit, visits per unit]; scripts/isa_phase_table.py --hits reads this
           "phases": {str(i): [w[i] / units, w[32 + i] / units] for i in range(32) if w[32 + i]},
           "phase_share_percent": {wn.get(i, str(i)): round(100.0 * w[i] / wtot, 1) for i in range(32) if w[32 + i]},
           "ticks_per_visit": {wn.get(i, str(i)): round(w[i] / w[32 + i]) for i in range(32) if w[32 + i]},
           "parse_ticks_per_unit": wtot / units,
           "entropy_ticks_per_unit": {names_e[i]: round(v[16 + i] / units) for i in range(8)},
           "entropy_extra": [round(v[16 + i] / units) for i in range(8, 12)],
           "phaseB_jobs_ticks_per_unit": {"huf_build_codes": round(v[28] / units), "huf other (mode, write table)": round(v[31] / units),
                                          "fse table build (sum of 3)": round(v[29] / units), "fse chains (sum of 3)": round(v[30] / units)}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
