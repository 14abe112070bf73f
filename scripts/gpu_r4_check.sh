#!/bin/bash
# scripts/gpu_r4_check.sh — one gpurun call: the decoder probe (1 and 40 units), smoke(), then the whole -m gpu suite, everything under its own deadline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 60 python scripts/decode_probe.py zstd_amd/libzstd_hip.so 1 2>&1 | tail -2
timeout -s KILL 60 python scripts/decode_probe.py zstd_amd/libzstd_hip.so 40 2>&1 | tail -2
timeout -s KILL 180 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -s KILL ${SUITE_TIMEOUT:-1200} python -m pytest tests -m gpu -q ${PYTEST_ARGS--x} 2>&1 | tail -60 | tee gpurun_out/pytest_gpu_full.log
