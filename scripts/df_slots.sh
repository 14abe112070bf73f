#!/bin/bash
# scripts/df_slots.sh — GPU box: level 3 (dfast) with persistent workgroups: how many table pairs should be live at once?
cd "$(dirname "$0")/.."
for cfg in "0:0" "4096:0" "2048:0" "1024:0" "512:0" "2048:18000" "1024:38000" "512:78000"; do
  S=${cfg%%:*}; P=${cfg#*:}
  ZHIP_DF_SLOTS=$S ZHIP_DF_LDS_PAD=$P timeout 300 python bench.py --level 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('slots $S pad $P', d['value'], 'MB/s parse', d['pipeline']['parse_ms'], d['parity']['bytes_identical_to_oracle_first_64_units'])"
done
