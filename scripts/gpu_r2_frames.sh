#!/bin/bash
# scripts/gpu_r2_frames.sh — one gpurun call: multi-block frame parity on the GPU, then frame timings
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frames.py tests/test_gpu_compress.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/pytest_gpu_frames.log
timeout 600 python scripts/frames_timing.py 2>&1 | tee gpurun_out/frames_timing.log
