#!/bin/bash
# scripts/pmc_frames.sh — run on the GPU box: rocprofv3 counter passes (own runs, kernel-trace only) over ONE job-pool frame of 1 GiB
# (scripts/frames_mt_timing.py, level $1, default job size unless $2): SQ counters, FETCH_SIZE, WRITE_SIZE, kernel stats.
LEVEL=${1:-1}; JOBSZ=${2:-0}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_frames_L$LEVEL
mkdir -p $OUT
export TMPDIR=/tmp LEVEL SIZE=$((1<<30)) KINDS=datagen JOBS=$JOBSZ
cd /tmp
B="python $ROOT/scripts/frames_mt_timing.py"
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -f csv -d $OUT/sq -o sq -- $B > $OUT/b_sq.json 2> $OUT/sq.err
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --kernel-trace -f csv -d $OUT/sq2 -o sq2 -- $B > $OUT/b_sq2.json 2> $OUT/sq2.err
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $OUT/fetch -o fetch -- $B > $OUT/b_f.json 2> $OUT/f.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $OUT/write -o write -- $B > $OUT/b_w.json 2> $OUT/w.err
timeout 200 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o stats -- $B > $OUT/b_s.json 2> $OUT/s.err
python $ROOT/scripts/pmc_summary.py $OUT | tee $OUT/summary.txt
