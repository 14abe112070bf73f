#!/bin/bash
# scripts/gpu_r4_lazy_pmc.sh — one gpurun call: FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, --kernel-trace only) of the level-5 unit kernels on 1 GiB of
# datagen (live rows, one parse): what the records, the links and the live rows cost in HBM traffic per launch
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/lazy_pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  ZHIP_ROW_MATCHER=enable timeout 150 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/$C -o p -- python $ROOT/bench.py --level 5 --mib 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --no-pipelined-extra > $OUT/$C.log 2>&1
done
python $ROOT/scripts/pmc_summary.py $OUT > $ROOT/gpurun_out/r04_L5_units_pmc.txt
cat $ROOT/gpurun_out/r04_L5_units_pmc.txt
