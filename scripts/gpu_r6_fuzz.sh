#!/bin/bash
# round 6: the big GPU fuzzers (units with explicit parameters, multi-block frames incl. dfast, lazy frames) on the round's kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/gpu_fuzz.log
: > $out
for s in 611 612 613; do timeout 900 python tests/tools/gpu_fuzz_units.py $s 120 2>&1 | tail -1 >> $out; done
for s in 81 82; do timeout 900 python tests/tools/gpu_fuzz_frames.py $s 40 2>&1 | tail -1 >> $out; done
timeout 900 python tests/tools/gpu_fuzz_lazy_frames.py 91 20 2>&1 | tail -1 >> $out
cat $out
