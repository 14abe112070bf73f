#!/bin/bash
# scripts/gpu_r4_lazy_sq.sh — one gpurun call: the SQ counter pass (own rocprofv3 --pmc run, --kernel-trace only) of the level-5 unit kernels on 1 GiB of datagen:
# how much of k_parse_lazy's wave-time is waiting (the "latency-bound, one round trip after the other" claim of DESIGN 4.2b in numbers)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/lazy_sq
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ZHIP_ROW_MATCHER=enable timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -f csv -d $OUT/sq -o sq -- python $ROOT/bench.py --level 5 --mib 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --no-pipelined-extra > $OUT/sq.log 2>&1
python $ROOT/scripts/pmc_summary.py $OUT > $ROOT/gpurun_out/r04_L5_units_sq.txt
cat $ROOT/gpurun_out/r04_L5_units_sq.txt
