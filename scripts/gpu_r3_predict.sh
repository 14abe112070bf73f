#!/bin/bash
# scripts/gpu_r3_predict.sh — one gpurun call: the row matcher's two-pass prediction: parity on the GPU, then level-5 benches with it on / off,
# then the lazy-strategy frames of 1 MiB with it on / off
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_rowhash.py tests/test_gpu_zz_decode_big.py tests/test_gpu_frames_lazy.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/pytest_gpu_predict.log
B="--level 5 --mib 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined-extra --no-extra-legs"
for W in datagen text; do for P in 1 0; do
  echo "== workload $W ZHIP_RH_PREDICT=$P" | tee -a gpurun_out/bench_L5_predict.log
  ZHIP_RH_PREDICT=$P timeout 150 python bench.py $B --workload $W 2>/dev/null | tail -1 | tee -a gpurun_out/bench_L5_predict.log
done; done
for P in 1 0; do
  echo "== frames ZHIP_LZ_PREDICT=$P" | tee -a gpurun_out/frames_lazy_predict.log
  ZHIP_LZ_PREDICT=$P LEVELS=5 NFRAMES=256 JOBPOOL_MIB=0 timeout 120 python scripts/frames_lazy_timing.py 2>/dev/null | tee -a gpurun_out/frames_lazy_predict.log
done
