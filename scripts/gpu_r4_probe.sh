#!/bin/bash
# scripts/gpu_r4_probe.sh — one gpurun call: lazy-strategy frames with the prediction behind its 32 KB probe ($ZHIP_LZ_PREDICT=1) and without it (=0):
# 256 x 1 MiB datagen / text frames, one 64 MiB job-pool frame (datagen), one 64 MiB job-pool frame of text
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r04_predict_probe.log
: > $L
for P in 1 0; do
  echo "== frames level 5 ZHIP_LZ_PREDICT=$P" | tee -a $L
  ZHIP_LZ_PREDICT=$P REPS=2 LEVELS=5 NFRAMES=256 JOBPOOL_MIB=${JOBPOOL_MIB:-64} timeout 240 python scripts/frames_lazy_timing.py 2>/dev/null | cut -c1-360 | tee -a $L
done
