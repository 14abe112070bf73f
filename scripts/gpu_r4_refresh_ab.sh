#!/bin/bash
# scripts/gpu_r4_refresh_ab.sh — one gpurun call: level-5 units on datagen (the bench's level5_row_prediction leg) with the product library and with the
# variant built without the batch refresh (scripts/build_variant.sh norefresh zhip_k_lazy -DZHIP_RH_REFRESH=0)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r04_refresh_ab.log
: > $L
for V in "" zstd_amd/variants/norefresh.so; do
  echo "== units level 5 datagen 256 MiB, library ${V:-product} (off / on = the two-pass prediction)" | tee -a $L
  ZHIP_LIB=${V:+$PWD/$V} timeout 150 python bench.py --leg level5_row_prediction --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('off','on','same_bytes','bytes_identical_to_oracle_first_8_units','error')}))" | tee -a $L
done
