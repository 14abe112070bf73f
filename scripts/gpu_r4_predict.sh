#!/bin/bash
# scripts/gpu_r4_predict.sh — one gpurun call: the two-pass prediction timed — level-5 units on text (datagen is a leg of the default line), then the
# lazy-strategy frames (256 x 1 MiB, one 64 MiB job-pool frame) with $ZHIP_LZ_PREDICT off and on
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
B="--level 5 --mib 512 --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined-extra --no-extra-legs --no-parity"
for W in text; do for P in 0 1; do
  echo "== units level 5, workload $W ZHIP_RH_PREDICT=$P" | tee -a gpurun_out/r04_L5_predict.log
  ZHIP_ROW_MATCHER=enable ZHIP_RH_PREDICT=$P timeout 150 python bench.py $B --workload $W 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step','pipeline','ratio')} | {'kernels_ms': d['roofline'].get('kernels_ms')}))" | tee -a gpurun_out/r04_L5_predict.log
done; done
for P in 0 1; do
  echo "== frames level 5 ZHIP_LZ_PREDICT=$P" | tee -a gpurun_out/r04_L5_predict.log
  ZHIP_LZ_PREDICT=$P LEVELS=5 NFRAMES=256 JOBPOOL_MIB=${JOBPOOL_MIB:-64} timeout 200 python scripts/frames_lazy_timing.py 2>/dev/null | cut -c1-330 | tee -a gpurun_out/r04_L5_predict.log
done
