#!/bin/bash
# round 5, last A/B: ds_mskor tag insert (ZSTD_fast, level 1) and the level-5 parser at five wavefronts per SIMD (smaller (row, tag) map) / max-ILP scheduling
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
out=gpurun_out/r05/ab_final.log
: > $out
run() { ZHIP_LIB=$1 timeout 200 python scripts/ab_parse.py $2 $3 1024 >> $out 2>&1; }
run "" 5 datagen,text
for v in lazy_b_maxilp lazy_c_f15o5 lazy_d_f14o6 lazy_e_f15o5_maxilp; do run $PWD/zstd_amd/variants/$v.so 5 datagen,text; done
run "" 1 datagen,silesia
run $PWD/zstd_amd/variants/fast_b_mskor.so 1 datagen,silesia
run "" 1 datagen,silesia
run $PWD/zstd_amd/variants/fast_b_mskor.so 1 datagen,silesia
grep '^{' $out
