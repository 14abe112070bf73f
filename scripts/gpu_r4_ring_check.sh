#!/bin/bash
# scripts/gpu_r4_ring_check.sh — one gpurun call: the GPU test files that exercise the lazy strategies (and the decoder's new many-frames test) first,
# then the live rows timed (scripts/gpu_r4_ring.sh)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_frames_lazy.py tests/test_gpu_rowhash.py tests/test_gpu_prediction.py tests/test_gpu_dropin_lazy.py tests/test_gpu_decode.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r04_ring_pytest.log
bash scripts/gpu_r4_ring.sh
