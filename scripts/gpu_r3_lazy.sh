#!/bin/bash
# scripts/gpu_r3_lazy.sh — one gpurun call: lazy-strategy frame parity on the GPU, then their timings under rocprofv3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frames_lazy.py tests/test_gpu_frames.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_frames_lazy.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lazy
mkdir -p $OUT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o lazy -- python $GRAFT_REPO_ROOT/scripts/frames_lazy_timing.py 2>$OUT/err.log | tee $GRAFT_REPO_ROOT/gpurun_out/frames_lazy_timing.log
find $OUT -name "*kernel_stats*.csv" -exec head -30 {} \; | tee $GRAFT_REPO_ROOT/gpurun_out/frames_lazy_rocprof_stats.txt
rm -f $OUT/*kernel_trace* $OUT/*/*kernel_trace* 2>/dev/null
