#!/bin/bash
# round 5: the plugin / multi tests and the default bench line
cd "$(dirname "$0")/.."
tag=${1:-a}
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_plugin.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r05/pytest_plugin_multi_$tag.log 2>&1
tail -15 gpurun_out/r05/pytest_plugin_multi_$tag.log
timeout 900 python bench.py > gpurun_out/r05/bench_default_$tag.json 2> gpurun_out/r05/bench_default_$tag.err
tail -c 2000 gpurun_out/r05/bench_default_$tag.json
tail -5 gpurun_out/r05/bench_default_$tag.err
