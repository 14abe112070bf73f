#!/bin/bash
# round 6, final kernels: the GPU fuzzers again with new seeds + the degenerate-shape fuzzer (tests/tools/gpu_fuzz_shapes.py)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/gpu_fuzz_final.log
: > $out
for s in ${FUZZ_SEEDS:-7001 7002 7003 7004}; do timeout 900 python tests/tools/gpu_fuzz_shapes.py $s 60 2>&1 | grep -i "mismatch\|seed\|error\|Traceback" | tail -6 >> $out; done
for s in 621 622; do timeout 900 python tests/tools/gpu_fuzz_units.py $s 120 2>&1 | tail -1 >> $out; done
for s in 83; do timeout 900 python tests/tools/gpu_fuzz_frames.py $s 40 2>&1 | tail -1 >> $out; done
timeout 900 python tests/tools/gpu_fuzz_lazy_frames.py 92 20 2>&1 | tail -1 >> $out
cat $out
