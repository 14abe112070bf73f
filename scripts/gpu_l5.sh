#!/bin/bash
# scripts/gpu_l5.sh — GPU box: levels 5 / 7 on datagen and text, row-hash matcher (the reference's default) vs hash chain
cd "$(dirname "$0")/.."
for L in 5 7; do
for W in "--workload datagen --mib 512" "--workload text --total-bytes 500000000"; do
for R in auto disable; do
  ZHIP_ROW_MATCHER=$R timeout 300 python bench.py --level $L --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs $W 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('L$L $W row=$R', d['value'], 'MB/s', d['pipeline'].get('parse_ms'), d['roofline'].get('kernels_ms'), d['parity'])"
done; done; done
