#!/bin/bash
# scripts/gpu_records.sh — GPU box: dictionary path parity + the records workload (BASELINE configs[4]) at level 3
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_dict.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --workload records --level 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('records L3', d['value'], 'MB/s', d['pipeline'])"
