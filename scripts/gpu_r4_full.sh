#!/bin/bash
# scripts/gpu_r4_full.sh — one gpurun call: what the driver runs at round end — the -m gpu suite, smoke(), the default bench line — each under a deadline
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout -s KILL ${SUITE_TIMEOUT:-1500} python -m pytest tests -m gpu -q -x 2>&1 | tail -25 ) 2>&1 | tee gpurun_out/pytest_gpu_full.log
( time timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/smoke.log
( time timeout -s KILL ${BENCH_TIMEOUT:-1200} python bench.py ${BENCH_ARGS} > gpurun_out/bench_default.jsonl 2> gpurun_out/bench_default.err ) 2>&1 | tail -4
tail -c 600 gpurun_out/bench_default.err
wc -l gpurun_out/bench_default.jsonl
tail -1 gpurun_out/bench_default.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','error','skipped','leg_wall_s','frac','achieved','avg_launch_ms')}) for k,v in d.items() if k not in ('config',)})
"
