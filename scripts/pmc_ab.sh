#!/bin/bash
# scripts/pmc_ab.sh <lib.so|-> <shape> [level] [mib] — GPU box, measurement helper: SQ counters of one library build on one input shape
LIB=$1; SHAPE=${2:-text}; LEVEL=${3:-1}; MIB=${4:-256}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$(basename ${LIB%.so})_$SHAPE
OUT=$ROOT/gpurun_out/pmc_ab/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
[ "$LIB" != "-" ] && export ZHIP_LIB=$ROOT/$LIB
cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -f csv -d $OUT/sq -o sq -- python $ROOT/scripts/ab_parse.py $LEVEL $SHAPE $MIB > $OUT/run.json 2> $OUT/sq.err
python $ROOT/scripts/pmc_summary.py $OUT | grep -A9 "k_parse" | tee $ROOT/gpurun_out/pmc_ab/summary_$TAG.txt
