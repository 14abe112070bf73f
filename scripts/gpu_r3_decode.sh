#!/bin/bash
# scripts/gpu_r3_decode.sh — one gpurun call: block-parallel decode of large frames: parity on the GPU, then the 1 GiB job-pool frame under rocprofv3
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_zz_decode_big.py tests/test_gpu_decode.py tests/test_gpu_frames_lazy.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_decode_big.log
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_decbig
mkdir -p $OUT
cd /tmp && SIZE=$((1<<30)) timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT -o dec -- python $GRAFT_REPO_ROOT/scripts/big_frame_decode.py 2>$OUT/err.log | tee $GRAFT_REPO_ROOT/gpurun_out/big_frame_decode.log
find $OUT -name "*kernel_stats*.csv" -exec head -24 {} \; | tee $GRAFT_REPO_ROOT/gpurun_out/big_frame_decode_rocprof_stats.txt
rm -f $OUT/*kernel_trace* $OUT/*/*kernel_trace* 2>/dev/null
