#!/bin/bash
# scripts/gpu_r4_frames_pmc.sh — one gpurun call: FETCH_SIZE / WRITE_SIZE / SQ passes (separate rocprofv3 --pmc runs, --kernel-trace only) of the lazy-strategy frame kernels:
# 256 datagen + 256 text-like frames of 1 MiB at level 5 (live rows, probed prediction)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/frames_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
  T=$(echo $C | cut -d' ' -f1)
  REPS=1 LEVELS=5 NFRAMES=256 JOBPOOL_MIB=0 timeout 150 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/$T -o p -- python $ROOT/scripts/frames_lazy_timing.py > $OUT/$T.log 2>&1
done
python - <<'PY' > $ROOT/gpurun_out/r04_frames_lazy_pmc.txt
import csv, glob, os, re
from collections import defaultdict
root = os.environ.get("OUT", "/root/repo/gpurun_out/frames_pmc")
acc = defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?").split("(")[0]
        if "zhip" in k:
            acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
print("# per dispatch, in launch order: [datagen frames, text-like frames] (FETCH_SIZE / WRITE_SIZE in KiB)")
for (k, c), v in sorted(acc.items()):
    print(f"{k:28s} {c:20s} " + "  ".join(f"{x:16.1f}" for x in v))
PY
cat $ROOT/gpurun_out/r04_frames_lazy_pmc.txt
