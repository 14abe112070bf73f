#!/bin/bash
# measurement helper: parse-kernel time vs resident units per CU (extra LDS per unit lowers residency)
for pad in 0 2560 5120 8960 14848; do
  ZHIP_PARSE_LDS_PAD=$pad python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.readline()); print('pad', os.environ.get('ZHIP_PARSE_LDS_PAD'), 'parse_ms', d['pipeline']['parse_ms'])"
done
