#!/bin/bash
# round 5: A/B of the dfast window's switches (variants built by scripts/build_variant.sh), level 3, three shapes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_dfast_${1:-x}.log
: > $out
for v in zstd_amd/variants/df_*.so; do
  ZHIP_LIB=$PWD/$v timeout 600 python scripts/ab_parse.py 3 silesia,text,datagen ${2:-2048} >> $out 2>&1
done
grep '^{' $out
