#!/usr/bin/env python3
"""scripts/pmc_summary.py <dir> — per-kernel averages of every counter found in the rocprofv3 counter_collection CSVs under
<dir>, plus the --stats kernel table, for the zhip kernels."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            if "zhip" not in k:
                continue
            a = acc[k][row.get("Counter_Name")]
            a[0] += 1; a[1] += float(row.get("Counter_Value", 0))
for k in sorted(acc):
    print(f"== {k}")
    for c, (n, v) in sorted(acc[k].items()):
        print(f"   {c:24s} dispatches={n:3d} avg/dispatch={v / n:18.1f}")
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("== rocprofv3 --kernel-trace --stats")
    for row in csv.reader(open(f)):
        if row and ("zhip" in row[0] or row[0] == "Name"):
            print("  ".join(c[:60] for c in row))
