#!/bin/bash
# scripts/gpu_r2_bench.sh — one gpurun call: plugin / decode / drop-in GPU tests, then the default bench line (as the driver runs it)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_plugin.py tests/test_gpu_decode.py tests/test_gpu_dropin.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu_b.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err ) 2>&1 | tail -3
tail -c 6000 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
