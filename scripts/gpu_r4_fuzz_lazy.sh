#!/bin/bash
# scripts/gpu_r4_fuzz_lazy.sh — one gpurun call: the lazy strategies fuzzed on the GPU after the live rows went in — units (gpu_fuzz_units.py: levels 1 .. 7 and explicit
# strategies, row matcher on / off) and frames / job-pool frames (gpu_fuzz_lazy_frames.py)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r04_fuzz_live_rows.log
: > $L
timeout 200 python tests/tools/gpu_fuzz_units.py ${SEED:-71} ${TRIALS:-120} 2>&1 | tail -5 | tee -a $L
timeout 200 python tests/tools/gpu_fuzz_lazy_frames.py ${SEED:-71} ${SECONDS_FRAMES:-90} 2>&1 | tail -8 | tee -a $L
