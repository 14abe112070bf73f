#!/bin/bash
# scripts/df_sweep.sh — GPU box: level-3 match-finder time for several dfast batch-width settings (kMul/8 << 4 | kAdd)
cd "$(dirname "$0")/.."
for W in 0 196 132 130 100 98 68; do
  ZHIP_DF_WIDTH=$W timeout 200 python bench.py --level 3 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('W=$W', d['value'], d['pipeline']['parse_ms'], d['parity']['bytes_identical_to_oracle_first_64_units'])"
done
