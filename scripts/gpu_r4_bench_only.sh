#!/bin/bash
# scripts/gpu_r4_bench_only.sh — one gpurun call: the default bench line alone (bench.py changed, the library did not)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout -s KILL ${BENCH_TIMEOUT:-900} python bench.py ${BENCH_ARGS} > gpurun_out/bench_default.jsonl 2> gpurun_out/bench_default.err ) 2>&1 | tail -4
tail -c 300 gpurun_out/bench_default.err
wc -l gpurun_out/bench_default.jsonl
tail -1 gpurun_out/bench_default.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('value','error','skipped','leg_wall_s','frac','achieved','avg_launch_ms','off','on','first_256MiB_off')}) for k,v in d.items() if k not in ('config',)})
"
