#!/bin/bash
# scripts/pipe_sweep.sh — GPU box: end-to-end rate of the default workload against the number of pipeline chunks
cd "$(dirname "$0")/.."
for C in 1 2 3 4 5 6 8; do
  ZHIP_PIPELINE_CHUNKS=$C timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --no-pipelined-extra 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('chunks $C', d['value'], 'MB/s', d['ms_per_step'], d['parity']['bytes_identical_to_oracle_first_64_units'])"
done
