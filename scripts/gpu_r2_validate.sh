#!/bin/bash
# scripts/gpu_r2_validate.sh — end-of-round check on the GPU box: the whole -m gpu suite, smoke(), the driver's default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6 | tee gpurun_out/r02/pytest_gpu_full.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r02/smoke.log
( time timeout 600 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.err ) 2>&1 | tail -3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r02/bench_default.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "ratio")})
print("roofline", d.get("roofline")); print("cpu", d.get("cpu_baseline"))
for k in ("end_to_end", "multi_block_frames", "job_pool_frame", "decode"):
    print(k, json.dumps(d.get(k))[:600])
print("silesia", json.dumps(d.get("silesia_shaped_level1"))[:700])
PY
