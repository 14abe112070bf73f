#!/bin/bash
# round 6: the records leg (BASELINE configs[4], reduced to N records) for the product library and every zstd_amd/variants/dict_*.so
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/records_${1:-x}.log
: > $out
for rep in 1 2; do
for lib in zstd_amd/libzstd_hip.so zstd_amd/variants/dict_*.so; do
  [ -f "$lib" ] || continue
  ZHIP_LIB=$PWD/$lib timeout 600 python bench.py --workload records --records ${2:-2000000} --base-records 500000 --level 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-pipelined-extra 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib', d['value'], 'MB/s', d['ms_per_step'], 'ms', d.get('pipeline'), d.get('parity', {}))
" >> $out
done
done
cat $out | cut -c1-330
