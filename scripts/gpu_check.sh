#!/bin/bash
# scripts/gpu_check.sh — one gpurun call: GPU parity tests, then short benches (each step under its own timeout)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.log
for L in "$@"; do
  timeout 300 python bench.py --level $L --steps 3 --warmup 1 > gpurun_out/bench_L$L.json 2> gpurun_out/bench_L$L.err
  echo "bench L$L rc=$?"; tail -c 3000 gpurun_out/bench_L$L.json; tail -3 gpurun_out/bench_L$L.err
done
