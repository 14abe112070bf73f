#!/bin/bash
# scripts/gpu_r4_lazy_stats.sh — one gpurun call: rocprofv3 --kernel-trace --stats of the lazy strategies as they run at the end of round 4 (live rows; frames with
# the probed prediction): level-5 units on 1 GiB of datagen (bench.py --level 5), then 256 datagen + 256 text frames of 1 MiB (scripts/frames_lazy_timing.py)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/lazy_stats
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ZHIP_ROW_MATCHER=enable timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/units -o s -- python $ROOT/bench.py --level 5 --mib 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --no-pipelined-extra > $OUT/units.log 2>&1
REPS=2 LEVELS=5 NFRAMES=256 JOBPOOL_MIB=0 timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/frames -o s -- python $ROOT/scripts/frames_lazy_timing.py > $OUT/frames.log 2>&1
L=$ROOT/gpurun_out/r04_lazy_kernel_stats.txt
: > $L
for t in units frames; do
  echo "#### $t: $(tail -1 $OUT/$t.log | cut -c1-400)" >> $L
  f=$(find $OUT/$t -name "*kernel_stats.csv" | head -1)
  echo "## $f" >> $L
  head -14 "$f" | cut -d, -f1-8 >> $L
done
cat $L
