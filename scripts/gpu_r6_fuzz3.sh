#!/bin/bash
# round 6: a long run of the degenerate-shape fuzzer alone: scripts/gpu_r6_fuzz3.sh FIRST_SEED COUNT [TRIALS]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/gpu_fuzz_shapes_$1.log
: > $out
for ((s=$1; s<$1+$2; s++)); do timeout 600 python tests/tools/gpu_fuzz_shapes.py $s ${3:-60} 2>&1 | grep -i "mismatch\|seed\|error\|Traceback" | tail -6 >> $out; done
grep -c "0 mismatches" $out; grep -i "mismatch trial\|error\|Traceback" $out | head -20
