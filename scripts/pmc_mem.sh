#!/bin/bash
# scripts/pmc_mem.sh <tag> <level> — GPU box: memory-path counters (TA / TCP / UTCL1 / TCC), one --pmc set per run
TAG=${1:-x}; LEVEL=${2:-3}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/mem_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $ROOT/bench.py --level $LEVEL --steps 2 --warmup 1 --no-cpu-baseline"
i=0
for SET in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace -f csv -d $OUT/p$i -o p$i -- $B > $OUT/b$i.json 2> $OUT/e$i.err || tail -3 $OUT/e$i.err
done
python $ROOT/scripts/pmc_summary.py $OUT | tee $OUT/summary_$TAG.txt
