#!/bin/bash
# round 6: rocprofv3 evidence for EVERY leg of the default bench line (round-5 verdict, item 6 i): per leg one --kernel-trace --stats pass, one FETCH_SIZE pass, one
# WRITE_SIZE pass and two SQ passes (separate --pmc runs with --kernel-trace only, as the guide prescribes); summaries -> gpurun_out/r06/prof/summary_<leg>.txt
# usage: scripts/gpu_r6_profiles.sh [leg ...]   (default: all)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r06/prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
SQ2="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
run() {  # tag, command...
  local TAG=$1; shift
  local D=$OUT/$TAG
  rm -rf $D; mkdir -p $D
  timeout 500 rocprofv3 --kernel-trace --stats -f csv -d $D/stats -o s -- "$@" > $D/stats.out 2> $D/stats.err || echo "$TAG stats: rc $?"
  timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $D/fetch -o p -- "$@" > $D/fetch.out 2> $D/fetch.err || echo "$TAG fetch: rc $?"
  timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $D/write -o p -- "$@" > $D/write.out 2> $D/write.err || echo "$TAG write: rc $?"
  timeout 500 rocprofv3 --pmc $SQ1 --kernel-trace -f csv -d $D/sq1 -o p -- "$@" > $D/sq1.out 2> $D/sq1.err || echo "$TAG sq1: rc $?"
  timeout 500 rocprofv3 --pmc $SQ2 --kernel-trace -f csv -d $D/sq2 -o p -- "$@" > $D/sq2.out 2> $D/sq2.err || echo "$TAG sq2: rc $?"
  { echo "# $TAG: $*"; echo "# rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc <SQ set 1> / --pmc <SQ set 2>: five separate runs (counter values: KiB per dispatch for the TCC sizes)";
    python3 $ROOT/scripts/pmc_summary.py $D; } > $OUT/summary_$TAG.txt
  rm -rf $D/stats $D/fetch $D/write $D/sq1 $D/sq2       # the raw CSVs stay on the box (tens of MB); the summary is what is kept
  echo "#### $TAG"; grep -c "^==" $OUT/summary_$TAG.txt
}
B="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-extra-legs --no-pipelined-extra"
L="python $ROOT/bench.py --no-cpu-baseline --steps 3 --leg"
want() { [ $# -eq 0 ] && return 0; }
LEGS_ALL="datagen_L1 text_L1 silesia4_L1 lorem_L1 silesia64_L3 records_L3 datagen_L5 frames_1MiB job_pool_1GiB plugin_B1 decode_L1"
LEGS=${@:-$LEGS_ALL}
for leg in $LEGS; do
  case $leg in
    datagen_L1)   run datagen_L1 $B --level 1 --mib 1024 ;;
    text_L1)      run text_L1 $B --workload text --total-bytes 1000000000 --level 1 ;;
    silesia4_L1)  run silesia4_L1 $B --workload silesia --copies 4 --level 1 ;;
    lorem_L1)     run lorem_L1 $B --workload lorem --mib 1024 --level 1 ;;
    silesia64_L3) run silesia64_L3 $B --workload silesia --copies 64 --level 3 ;;
    records_L3)   run records_L3 $B --workload records --records 10000000 --base-records 1000000 --level 3 ;;
    datagen_L5)   run datagen_L5 $B --level 5 --mib 1024 ;;
    frames_1MiB)  run frames_1MiB $L multi_block_frames ;;
    job_pool_1GiB) run job_pool_1GiB $L job_pool_frame ;;
    plugin_B1)    run plugin_B1 python $ROOT/scripts/plugin_prepare_only.py 1024 ;;     # the leg's device part (rocprofv3 crashes under the leg's 64 reference threads)
    decode_L1)    run decode_L1 $B --level 1 --mib 1024 --mode decode ;;
  esac
done
