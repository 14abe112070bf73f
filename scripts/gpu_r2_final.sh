#!/bin/bash
# scripts/gpu_r2_final.sh — round-2 measurement pass on the GPU box: full -m gpu suite, the driver's default bench line, the
# other named workloads, and the rocprofv3 kernel statistics of the default command (summaries are copied to profiles/ by hand)
cd "$(dirname "$0")/.."
ROOT=$(pwd)
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6 | tee gpurun_out/r02/pytest_gpu_full.log
( time timeout 900 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err ) 2>&1 | tail -3
for cfg in "text_L1:--level 1 --workload text --total-bytes 1000000000" "silesia8_L1:--level 1 --workload silesia --copies 8" \
           "datagen_L3:--level 3" "silesia8_L3:--level 3 --workload silesia --copies 8" "datagen_L5_row:--level 5" "records_L3:--level 3 --workload records" \
           "decode_datagen_L1:--mode decode"; do
  name=${cfg%%:*}; args=${cfg#*:}
  timeout 600 python bench.py --steps 10 --warmup 2 --no-extra-legs $args > gpurun_out/r02/bench_$name.json 2> gpurun_out/r02/bench_$name.err
  python - "$name" gpurun_out/r02/bench_$name.json <<'PY'
import sys, json
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    print(sys.argv[1], d["value"], d["unit"], "ratio", d.get("ratio"), "roofline", d.get("roofline", {}).get("frac"), "cpu", d.get("cpu_baseline", {}).get("value"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/gpurun_out/r02/stats -o stats -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined-extra --no-extra-legs > $ROOT/gpurun_out/r02/bench_stats.json 2> $ROOT/gpurun_out/r02/stats.err
python $ROOT/scripts/pmc_summary.py $ROOT/gpurun_out/r02 | tail -12 | tee $ROOT/gpurun_out/r02/rocprof_stats_default.txt
