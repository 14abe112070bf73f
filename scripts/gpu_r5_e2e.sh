#!/bin/bash
# round 5: host-buffer path with the staging copies split over host threads: lanes x chunk size at 1 GiB, and the 4 GiB figure
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
out=gpurun_out/r05/e2e_${1:-x}.log
: > $out
LANES=2,3,4 CUS=1024,1536 timeout 600 python scripts/e2e_sweep.py 2>/dev/null >> $out
MIB=4096 LANES=2,3 CUS=1024 timeout 600 python scripts/e2e_sweep.py 2>/dev/null >> $out
cat $out
