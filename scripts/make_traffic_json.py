#!/usr/bin/env python3
"""scripts/make_traffic_json.py — writes profiles/latest_traffic.json (what bench.py's `roofline.traffic` quotes) from the round's committed counter
summaries: profiles/r05_L1_datagen_sq_tcc.txt (scripts/pmc_sq.sh) and profiles/r05_pmc_legs_<leg>.txt (scripts/pmc_legs.sh).  FETCH_SIZE / WRITE_SIZE are KiB
per dispatch.  Note for the queue stages: rocprofv3 serialises kernels while it collects counters, so the LDS-table kernel (k_parse_fast_q, k_parse_dict_q)
takes every unit of the batch and its global-table co-kernel finds the queue empty — the figures are the STAGE's traffic with all units on the LDS form."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


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


def hbm(c):      # FETCH_SIZE x2 = the guide's gfx950 correction (calibrated on k_gather's coalesced reads), WRITE_SIZE as counted
    return int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)


head = counters(os.path.join(P, "r05_L1_datagen_sq_tcc.txt"))
out = {"source": "profiles/r05_L1_datagen_sq_tcc.txt (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only; scripts/pmc_sq.sh r05_L1_datagen 1 1024: "
                 "bench.py --level 1 --mib 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined-extra --no-extra-legs); counter passes serialise kernels: k_parse_fast_q takes all "
                 "8 192 units, the figure is the whole ZSTD_fast stage on the LDS form",
       "units": "counter values are KiB per dispatch, averaged over the dispatches of the run",
       "calibration": "k_gather reads exactly the compressed stream with 16 B/lane coalesced loads and reports FETCH_SIZE = 0.50x of it -> the gfx950 x2 correction of MI355X_MICROARCH.md "
                      "is applied to FETCH_SIZE; WRITE_SIZE of k_gather equals the bytes written, no correction.  For the match finders' 4-8 byte gathers the x2 is an upper bound: "
                      "both figures are kept (..._uncorrected)"}
for k in ("k_parse_fast_q", "k_entropy", "k_gather", "k_decode"):
    out[k] = {"FETCH_SIZE_KiB_raw": head[k]["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": head[k]["WRITE_SIZE"]}
out["k_parse_fast_hbm_bytes_per_launch"], out["k_parse_fast_hbm_bytes_per_launch_uncorrected"] = hbm(head["k_parse_fast_q"])
out["k_decode_hbm_bytes_per_launch"] = hbm(head["k_decode"])[0]
out["legs"] = {}
for leg, kern, key in (("silesia4_level1", "k_parse_fast_q", "k_parse_fast"), ("silesia64_level3", "k_parse_dfast", "k_parse_dfast"), ("records_zdict_level3", "k_parse_dict_q", "k_parse_dict")):
    c = counters(os.path.join(P, f"r05_pmc_legs_{leg}.txt"))[kern]
    a, b = hbm(c)
    out["legs"][leg] = {f"{key}_FETCH_SIZE_KiB_raw": c["FETCH_SIZE"], f"{key}_WRITE_SIZE_KiB_raw": c["WRITE_SIZE"], f"{key}_hbm_bytes_per_launch": a,
                        f"{key}_hbm_bytes_per_launch_uncorrected": b, "kernel_counted": kern,
                        "source": f"profiles/r05_pmc_legs_{leg}.txt (scripts/pmc_legs.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only, of bench.py with "
                                  f"this leg's workload flags, round 5; FETCH_SIZE x2 = the guide's gfx950 correction, WRITE_SIZE as counted; counter passes serialise kernels, so for the queue stages "
                                  f"(k_parse_fast_q/_g, k_parse_dict_q/_g) the LDS-table kernel takes every unit and the figure is the stage with all units on that form — not the mix the timed run executes)"}
# level 5: the match-finder stage is three kernels (link / list builder, record search, parse); the leg's figure is their sum
c5 = counters(os.path.join(P, "r05_L5_datagen_final_sq_tcc.txt"))
a5 = sum(hbm(c5[k])[0] for k in ("k_hc_chain", "k_hc_search_lds", "k_parse_lazy")); b5 = sum(hbm(c5[k])[1] for k in ("k_hc_chain", "k_hc_search_lds", "k_parse_lazy"))
out["legs"]["datagen_level5"] = {"k_parse_lazy_hbm_bytes_per_launch": a5, "k_parse_lazy_hbm_bytes_per_launch_uncorrected": b5, "kernel_counted": "k_hc_chain + k_hc_search_lds + k_parse_lazy",
                               "per_kernel_KiB_raw": {k: {"FETCH_SIZE": c5[k]["FETCH_SIZE"], "WRITE_SIZE": c5[k]["WRITE_SIZE"]} for k in ("k_hc_chain", "k_hc_search_lds", "k_parse_lazy")},
                               "source": "profiles/r05_L5_datagen_final_sq_tcc.txt (scripts/pmc_sq.sh r05_L5_final 5 1024, the kernels as shipped at the end of round 5: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only, of "
                                         "bench.py --level 5 --mib 1024; FETCH_SIZE x2 = the guide's gfx950 correction, WRITE_SIZE as counted; sum of the stage's three kernels)"}
json.dump(out, open(os.path.join(P, "latest_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "bytes_per_launch" in k}, indent=1))
print(json.dumps({l: {k: v for k, v in d.items() if "bytes_per_launch" in k} for l, d in out["legs"].items()}, indent=1))
