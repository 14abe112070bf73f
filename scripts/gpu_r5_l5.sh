#!/bin/bash
# round 5: level 5 (greedy, row matcher) on datagen 1 GiB + text: stage times and SHA, product build vs variants
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_l5_${1:-x}.log
: > $out
for v in zstd_amd/libzstd_hip.so zstd_amd/variants/l5_*.so; do
  [ -f $v ] || continue
  ROW=0 ZHIP_LIB=$PWD/$v timeout 600 python scripts/ab_parse.py 5 datagen,text 1024 >> $out 2>&1
done
grep '^{' $out
