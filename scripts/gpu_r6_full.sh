#!/bin/bash
# round 5: the whole -m gpu suite, smoke(), and the default bench line
cd "$(dirname "$0")/.."
tag=${1:-a}
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r06/pytest_gpu_full_$tag.log 2>&1
tail -6 gpurun_out/r06/pytest_gpu_full_$tag.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06/smoke_$tag.log 2>&1
tail -2 gpurun_out/r06/smoke_$tag.log
if [ "$2" != "nobench" ]; then
timeout 900 python bench.py > gpurun_out/r06/bench_default_$tag.json 2> gpurun_out/r06/bench_default_$tag.err
tail -c 1800 gpurun_out/r06/bench_default_$tag.json
fi
