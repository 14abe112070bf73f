#!/bin/bash
# scripts/gpu_r4_bigframe.sh — one gpurun call: ONE 1 GiB job-pool frame through the block-parallel decoder under rocprofv3: kernel stats, then FETCH_SIZE / WRITE_SIZE
# passes (separate --pmc runs, --kernel-trace only) of the k_bf_* kernels
cd "$(dirname "$0")/.."
R=$PWD; OUT=$R/gpurun_out/prof_bigframe; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
SIZE=$((1<<30)) timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o s -- python $R/scripts/big_frame_decode.py 2>$OUT/err.log | tee $OUT/run.log
for C in FETCH_SIZE WRITE_SIZE; do SIZE=$((1<<30)) timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/$C -o p -- python $R/scripts/big_frame_decode.py > /dev/null 2>&1; done
python $R/scripts/pmc_summary.py $OUT | grep -v "k_frame\|k_offsets\|k_gather" | tee $R/gpurun_out/r04_big_frame_decode_1GiB.txt
rm -rf $OUT/*/*kernel_trace* $OUT/*/*/*kernel_trace* 2>/dev/null
