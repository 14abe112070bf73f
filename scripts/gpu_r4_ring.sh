#!/bin/bash
# scripts/gpu_r4_ring.sh — one gpurun call: the row matcher's live rows (LzRing / rh_live_ring) timed against the walk through the links ($ZHIP_LZ_RING=0):
# level-5 units on datagen (the bench's level5_row_prediction leg: prediction off and on in one run), then the lazy-strategy frames (256 x 1 MiB datagen
# and text, one 64 MiB job-pool frame) with the prediction off and on.  The frames' ring-off times are in profiles/r04_L5_predict.log.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r04_live_rows.log
: > $L
for R in 1 0; do
  echo "== units level 5 datagen 256 MiB, ZHIP_LZ_RING=$R (off / on = the two-pass prediction)" | tee -a $L
  ZHIP_LZ_RING=$R timeout 150 python bench.py --leg level5_row_prediction --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('off','on','same_bytes','bytes_identical_to_oracle_first_8_units','error')}))" | tee -a $L
done
echo "== units level 5 datagen, ZHIP_LZ_RING=1, try budget 4096 instead of 256" | tee -a $L
ZHIP_RH_BUDGET=4096 timeout 150 python bench.py --leg level5_row_prediction --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('on','same_bytes','error')}))" | tee -a $L
for P in 0 1; do
  echo "== frames level 5 ZHIP_LZ_RING=1 ZHIP_LZ_PREDICT=$P" | tee -a $L
  ZHIP_LZ_PREDICT=$P REPS=2 LEVELS=5 NFRAMES=256 JOBPOOL_MIB=${JOBPOOL_MIB:-64} timeout 240 python scripts/frames_lazy_timing.py 2>/dev/null | cut -c1-360 | tee -a $L
done
echo "== frames level 5 ZHIP_LZ_RING=0 ZHIP_LZ_PREDICT=0 (256 x 1 MiB only: the digests to compare)" | tee -a $L
ZHIP_LZ_RING=0 ZHIP_LZ_PREDICT=0 REPS=1 LEVELS=5 NFRAMES=256 JOBPOOL_MIB=0 timeout 120 python scripts/frames_lazy_timing.py 2>/dev/null | cut -c1-360 | tee -a $L
