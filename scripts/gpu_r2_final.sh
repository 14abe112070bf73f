#!/bin/bash
# scripts/gpu_r2_final.sh — round-2 measurement pass on the GPU box: full -m gpu suite, smoke(), the driver's default bench line, the
# other named workloads, and the rocprofv3 kernel statistics of the default command (summaries are copied to profiles/ by hand)
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
( time timeout 300 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6 | tee gpurun_out/r02/pytest_gpu_full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/r02/smoke.log
( time timeout 400 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err ) 2>&1 | tail -3
for cfg in "text_L1:--level 1 --workload text --total-bytes 1000000000" "silesia8_L1:--level 1 --workload silesia --copies 8" \
           "datagen_L3:--level 3" "silesia8_L3:--level 3 --workload silesia --copies 8" "datagen_L5_row:--level 5" "records_L3:--level 3 --workload records" \
           "decode_datagen_L1:--mode decode"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 200 python bench.py --steps 10 --warmup 2 --no-extra-legs $args > gpurun_out/r02/bench_$name.json 2> gpurun_out/r02/bench_$name.err
  python - "$name" gpurun_out/r02/bench_$name.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["unit"], "ratio", d.get("ratio"), "roofline", d.get("roofline", {}).get("frac"), "cpu", d.get("cpu_baseline", {}).get("value"), "all", d.get("cpu_baseline", {}).get("all_cores", {}).get("value"), "parity", json.dumps(d.get("parity"))[:160])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/r02/stats -o stats -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined-extra --no-extra-legs > $ROOT/gpurun_out/r02/bench_stats.json 2> $ROOT/gpurun_out/r02/stats.err
python $ROOT/scripts/pmc_summary.py $ROOT/gpurun_out/r02 | tail -12 | tee $ROOT/gpurun_out/r02/rocprof_stats_default.txt
python - <<'PY'
import json
d = json.loads(open("/root/repo/gpurun_out/r02/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "ratio")}, "roofline", d.get("roofline", {}).get("frac"))
for k in ("end_to_end", "multi_block_frames", "job_pool_frame"):
    v = d.get(k) or {}
    print(k, v.get("value"), v.get("unit"), json.dumps(v.get("parity"))[:200], v.get("cpu_reference"))
print("decode", (d.get("decode") or {}).get("value"), "silesia", (d.get("silesia_shaped_level1") or {}).get("value"))
PY
