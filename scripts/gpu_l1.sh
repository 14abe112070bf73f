#!/bin/bash
# scripts/gpu_l1.sh — GPU box: level-1 match finder on the three input shapes (datagen P50, text, Silesia-shaped mix)
cd "$(dirname "$0")/.."
for W in "--workload datagen" "--workload text --total-bytes 1000000000" "--workload silesia --copies 4"; do
  timeout 300 python bench.py --level 1 --steps 3 --warmup 1 --no-cpu-baseline $W 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$W', d['value'], 'MB/s parse', d['pipeline']['parse_ms'], 'entropy', d['pipeline']['entropy_ms'], d['parity']['bytes_identical_to_oracle_first_64_units'])"
done
