#!/bin/bash
# scripts/gpu_r4_lazy_occ.sh — one gpurun call: level-5 units on 1 GiB of datagen (8 192 units: more than one round of resident wavefronts) with k_parse_lazy
# compiled for 5 / 6 / 8 wavefronts per SIMD (scripts/build_variant.sh lazyoccN zhip_k_lazy -DZHIP_LAZY_OCC=...) against the product's own choice
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r04_lazy_occ.log
: > $L
for V in "" zstd_amd/variants/lazyocc5.so zstd_amd/variants/lazyocc6.so zstd_amd/variants/lazyocc8.so; do
  for M in ${MIBS:-1024 256}; do
  echo "== units level 5 datagen $M MiB, library ${V:-product}" | tee -a $L
  ZHIP_L5_LEG_MIB=$M ZHIP_LIB=${V:+$PWD/$V} timeout 150 python bench.py --leg level5_row_prediction --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ('off','on','same_bytes','bytes_identical_to_oracle_first_8_units','error')}))" | tee -a $L
  done
done
