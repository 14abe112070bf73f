#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_entropy_${1:-x}.log
: > $out
for rep in 1 2; do for v in zstd_amd/variants/ent_*.so; do
  ZHIP_LIB=$PWD/$v timeout 600 python scripts/ab_parse.py 1 silesia,text,datagen 1024 >> $out 2>&1
done; done
grep '^{' $out
