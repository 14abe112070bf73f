#!/bin/bash
# round 6: the ZSTD_fast window's phase ticks / visits (PROF build) on the three level-1 shapes + A/B of the variants in zstd_amd/variants/fast_*.so
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
tag=${1:-a}
for w in datagen silesia text; do
  WORKLOAD=$w LEVEL=1 MIB=1024 timeout 600 python scripts/prof_phases.py > gpurun_out/r06/phases_${w}_L1_$tag.json 2> gpurun_out/r06/phases_${w}_L1_$tag.err
done
out=gpurun_out/r06/ab_fast_$tag.log
: > $out
for rep in 1 2; do
  timeout 600 python scripts/ab_parse.py 1 silesia,text,datagen 1024 >> $out 2>&1
  for v in zstd_amd/variants/fast_*.so; do
    [ -f "$v" ] && ZHIP_LIB=$PWD/$v timeout 600 python scripts/ab_parse.py 1 silesia,text,datagen 1024 >> $out 2>&1
  done
done
grep '^{' $out
