#!/bin/bash
# scripts/gpu_r2_check.sh — one gpurun call: level-1 GPU parity tests, then the three level-1 input shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parse.py tests/test_gpu_compress.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_l1.log
bash scripts/gpu_l1.sh 2>&1 | tee gpurun_out/l1_shapes.log
