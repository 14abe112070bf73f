#!/bin/bash
# scripts/pmc_sq.sh <tag> <level> [mib] — run on the GPU box: one rocprofv3 --pmc pass of SQ counters (own run, kernel-trace only)
# + one FETCH_SIZE pass + one WRITE_SIZE pass over a short bench.py run; prints per-kernel averages of the zhip kernels.
TAG=${1:-x}; LEVEL=${2:-3}; MIB=${3:-1024}; EXTRA="${4:-}"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --level $LEVEL --mib $MIB --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined-extra --no-extra-legs $EXTRA"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -f csv -d $OUT/sq -o sq -- $B > $OUT/b_sq.json 2> $OUT/sq.err
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --kernel-trace -f csv -d $OUT/sq2 -o sq2 -- $B > $OUT/b_sq2.json 2> $OUT/sq2.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/fetch -o fetch -- $B > $OUT/b_f.json 2> $OUT/f.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/write -o write -- $B > $OUT/b_w.json 2> $OUT/w.err
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- $B > $OUT/b_s.json 2> $OUT/s.err
python $ROOT/scripts/pmc_summary.py $OUT | tee $OUT/summary_$TAG.txt
