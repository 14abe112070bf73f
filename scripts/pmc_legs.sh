#!/bin/bash
# scripts/pmc_legs.sh — GPU box: FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, kernel-trace only) + kernel stats of the bench legs
# that are not the headline: Silesia-shaped x4 level 1, Silesia-shaped x64 level 3 (BASELINE configs[2]), 10 M records + dictionary (configs[4]).
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_legs
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() {  # tag, bench args...
  local TAG=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $C --kernel-trace -f csv -d $OUT/$TAG/$C -o p -- python $ROOT/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --no-pipelined-extra > $OUT/$TAG.$C.json 2> $OUT/$TAG.$C.err || tail -2 $OUT/$TAG.$C.err
  done
  timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $OUT/$TAG/stats -o s -- python $ROOT/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --no-pipelined-extra > $OUT/$TAG.stats.json 2> $OUT/$TAG.stats.err
  python $ROOT/scripts/pmc_summary.py $OUT/$TAG > $OUT/summary_$TAG.txt
}
run silesia4_level1 --workload silesia --copies 4 --level 1
run silesia64_level3 --workload silesia --copies 64 --level 3
run records_zdict_level3 --workload records --records 10000000 --base-records 1000000 --level 3
for t in silesia4_level1 silesia64_level3 records_zdict_level3; do echo "#### $t"; cat $OUT/summary_$t.txt; done
