#!/bin/bash
# round 5: dfast A/B — parity tests of the dfast paths, the level-3 bench legs, the phase split
cd "$(dirname "$0")/.."
tag=${1:-a}
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_compress.py tests/test_gpu_frames.py tests/test_gpu_parse.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/r05/pytest_dfast_$tag.log 2>&1
tail -3 gpurun_out/r05/pytest_dfast_$tag.log
for cfg in "silesia 3 16" "datagen 3 1" "text 3 1"; do
  set -- $cfg
  timeout 600 python bench.py --workload $1 --level $2 --copies $3 --mib 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined-extra > gpurun_out/r05/bench_${1}_L${2}_$tag.json 2> gpurun_out/r05/bench_${1}_L${2}_$tag.err
  python - <<PY
import json
try:
    j = json.loads(open("gpurun_out/r05/bench_${1}_L${2}_$tag.json").read().strip().splitlines()[-1])
    print("$1 L$2", j["value"], "MB/s parse_ms", j["pipeline"]["parse_ms"], "ratio", j["ratio"], "parity", j["parity"].get("full_size", {}).get("sha256_equals_reference_stream"), j["parity"].get("bytes_identical_to_oracle_first_64_units"))
except Exception as e:
    print("$1 L$2 failed", e)
PY
done
echo "== WORKLOAD=silesia LEVEL=3 MIB=4096 ($tag)" >> gpurun_out/r05/phases_$tag.log
WORKLOAD=silesia LEVEL=3 MIB=4096 timeout 300 python scripts/prof_phases.py >> gpurun_out/r05/phases_$tag.log 2>&1
grep -A14 dfast_ticks gpurun_out/r05/phases_$tag.log | tail -16
