#!/usr/bin/env python3
"""scripts/profile_summary.py <dir> <tag> — condense the rocprofv3 CSVs written by profile_round.sh into
gpurun_out/prof_<tag>/summary_<tag>.txt (kernel stats + HBM counters per dispatch) and traffic_<tag>.json."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, tag = sys.argv[1], sys.argv[2]
lines = []


def find(sub, pat):
    g = glob.glob(os.path.join(root, sub, "**", pat), recursive=True)
    return g[0] if g else None


st = find("stats", "*kernel_stats.csv")
if st:
    lines.append("== rocprofv3 --kernel-trace --stats (bench.py --steps 5 --warmup 2) ==")
    with open(st) as f:
        for row in csv.reader(f):
            lines.append("  ".join(c[:72] for c in row))
traffic = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    cc = find(sub, "*counter_collection.csv")
    if not cc:
        lines.append(f"({ctr}: no counter_collection.csv)")
        continue
    acc = defaultdict(lambda: [0, 0.0])
    with open(cc) as f:
        rd = csv.DictReader(f)
        for row in rd:
            if row.get("Counter_Name") != ctr:
                continue
            k = row.get("Kernel_Name", "?").split("(")[0]
            acc[k][0] += 1
            acc[k][1] += float(row.get("Counter_Value", 0))
    lines.append(f"== rocprofv3 --pmc {ctr} (bench.py --steps 2 --warmup 1): average per dispatch, raw counter units (KB) ==")
    for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k[:60]:60s} dispatches={n:4d} avg={v / n:14.1f}")
        traffic.setdefault(k, {})[ctr + "_KB_per_dispatch"] = v / n
open(os.path.join(root, f"summary_{tag}.txt"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(root, f"traffic_{tag}.json"), "w"), indent=1)
print("\n".join(lines))
