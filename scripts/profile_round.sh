#!/bin/bash
# scripts/profile_round.sh <tag> — run on the GPU box (via gpurun): rocprofv3 kernel stats + HBM traffic counters of bench.py.
# Counters are collected in their own passes (one --pmc set per run) with --kernel-trace only, as MI355X_MICROARCH.md prescribes.
TAG=${1:-r01}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined-extra"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- $BENCH > $OUT/bench_stats.json 2> $OUT/stats.err
BENCH2="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined-extra"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/fetch -o fetch -- $BENCH2 > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/write -o write -- $BENCH2 > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
python $ROOT/scripts/profile_summary.py $OUT $TAG
