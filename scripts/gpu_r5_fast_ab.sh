#!/bin/bash
# round 5: A/B of ZSTD_fast window variants (zstd_amd/variants/fast_*.so), level 1, three shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_fast_${1:-x}.log
: > $out
for rep in 1 2; do
for v in zstd_amd/variants/fast_*.so; do
  ZHIP_LIB=$PWD/$v timeout 600 python scripts/ab_parse.py 1 silesia,text,datagen ${2:-1024} >> $out 2>&1
done
done
grep '^{' $out
