#!/usr/bin/env python3
"""scripts/make_traffic_json.py — writes profiles/latest_traffic.json (what bench.py's `roofline.traffic` quotes, for EVERY leg of the default line) from the round's
committed counter summaries profiles/r06_prof_<leg>.txt (scripts/gpu_r6_profiles.sh: per leg separate rocprofv3 runs --kernel-trace --stats / --pmc FETCH_SIZE /
--pmc WRITE_SIZE / two SQ sets).  FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the x2 on reads is the gfx950 correction
of MI355X_MICROARCH.md, calibrated here on k_gather (coalesced 16 B/lane reads of exactly the compressed stream: FETCH_SIZE reports 0.50 x of it); for the match finders'
4-8 byte gathers it is an upper bound, so the uncorrected figure is kept beside it.
Note for the queue stages: rocprofv3 serialises kernels while it collects counters, so the LDS-table kernel (k_parse_fast_q, k_parse_dict_q) takes every unit of the batch
and its global-table co-kernel finds the queue empty — those figures are the STAGE's traffic with all units on the LDS form, not the mix a timed run executes."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
ROUND = "r06"


def counters(path):
    res, k = {}, None
    for l in open(path):
        m = re.match(r"== zhip::(\w+)", l)
        if m:
            k = m.group(1); res.setdefault(k, {}); continue
        m = re.match(r"\s+(\w+)\s+dispatches=\s*\d+\s+avg/dispatch=\s*([\d.]+)", l)
        if m and k:
            res[k][m.group(1)] = float(m.group(2))
    return res


def hbm(c):
    return int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)


def src(leg, what):
    return (f"profiles/{ROUND}_prof_{leg}.txt (scripts/gpu_r6_profiles.sh {leg}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only, of {what}, round 6; "
            "FETCH_SIZE x2 = the guide's gfx950 correction, WRITE_SIZE as counted; counter passes serialise kernels: for the queue stages the LDS-table kernel takes every unit)")


head = counters(os.path.join(P, f"{ROUND}_prof_datagen_L1.txt"))
out = {"source": src("datagen_L1", "bench.py --level 1 --mib 1024 --steps 2 --warmup 1 (the headline's configuration)"),
       "units": "counter values are KiB per dispatch, averaged over the dispatches of the run",
       "calibration": "k_gather reads exactly the compressed stream with 16 B/lane coalesced loads and reports FETCH_SIZE = 0.50x of it -> the gfx950 x2 correction of MI355X_MICROARCH.md "
                      "is applied to FETCH_SIZE; WRITE_SIZE of k_gather equals the bytes written, no correction.  For the match finders' 4-8 byte gathers the x2 is an upper bound: "
                      "both figures are kept (..._uncorrected)"}
for k in ("k_parse_fast_q", "k_entropy", "k_gather"):
    out[k] = {"FETCH_SIZE_KiB_raw": head[k]["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": head[k]["WRITE_SIZE"]}
out["k_parse_fast_hbm_bytes_per_launch"], out["k_parse_fast_hbm_bytes_per_launch_uncorrected"] = hbm(head["k_parse_fast_q"])
dec = counters(os.path.join(P, f"{ROUND}_prof_decode_L1.txt"))
out["k_decode"] = {"FETCH_SIZE_KiB_raw": dec["k_decode"]["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": dec["k_decode"]["WRITE_SIZE"]}
out["k_decode_hbm_bytes_per_launch"] = hbm(dec["k_decode"])[0]
out["legs"] = {}


def leg(name, file_leg, kernels, key, what):
    c = counters(os.path.join(P, f"{ROUND}_prof_{file_leg}.txt"))
    a = sum(hbm(c[k])[0] for k in kernels); b = sum(hbm(c[k])[1] for k in kernels)
    e = {f"{key}_hbm_bytes_per_launch": a, f"{key}_hbm_bytes_per_launch_uncorrected": b, "kernel_counted": " + ".join(kernels),
         "per_kernel_KiB_raw": {k: {"FETCH_SIZE": c[k]["FETCH_SIZE"], "WRITE_SIZE": c[k]["WRITE_SIZE"]} for k in kernels}, "source": src(file_leg, what)}
    out["legs"][name] = e


leg("silesia4_level1", "silesia4_L1", ["k_parse_fast_q"], "k_parse_fast", "bench.py --workload silesia --copies 4 --level 1")
leg("text_level1", "text_L1", ["k_parse_fast_q"], "k_parse_fast", "bench.py --workload text --total-bytes 1000000000 --level 1 (BASELINE configs[3] as one shard)")
leg("lorem_level1", "lorem_L1", ["k_parse_fast_q"], "k_parse_fast", "bench.py --workload lorem --mib 1024 --level 1")
leg("silesia64_level3", "silesia64_L3", ["k_parse_dfast"], "k_parse_dfast", "bench.py --workload silesia --copies 64 --level 3 (BASELINE configs[2])")
leg("records_zdict_level3", "records_L3", ["k_parse_dict_q"], "k_parse_dict", "bench.py --workload records --records 10000000 --base-records 1000000 --level 3 (BASELINE configs[4])")
leg("datagen_level5", "datagen_L5", ["k_hc_chain", "k_hc_search_lds", "k_parse_lazy"], "k_parse_lazy", "bench.py --level 5 --mib 1024 (sum of the stage's three kernels)")
leg("multi_block_frames", "frames_1MiB", ["k_frame_hbm"], "k_frame_hbm", "bench.py --leg multi_block_frames (1 024 frames of 1 MiB: more workgroups than the LDS-table form holds, so the all-HBM form runs)")
leg("job_pool_frame", "job_pool_1GiB", ["k_frame_fast"], "k_frame_fast", "bench.py --leg job_pool_frame (one 1 GiB frame, job table)")
leg("plugin_B1", "plugin_B1", ["k_parse_fast_q"], "k_parse_fast", "scripts/plugin_prepare_only.py 1024 (the leg's device part: zhip_prepare_sequences on 16 384 blocks of 64 KB; rocprofv3 crashes under the leg's 64 reference threads)")
json.dump(out, open(os.path.join(P, "latest_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "bytes_per_launch" in k}, indent=1))
print(json.dumps({l: {k: v for k, v in d.items() if "bytes_per_launch" in k} for l, d in out["legs"].items()}, indent=1))
