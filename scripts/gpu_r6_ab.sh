#!/bin/bash
# round 6: A/B of the product library against every zstd_amd/variants/*.so: scripts/gpu_r6_ab.sh TAG [level] [shapes] [MiB]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_${1:-x}.log
: > $out
for rep in 1 2; do
  timeout 600 python scripts/ab_parse.py ${2:-1} ${3:-silesia,text,datagen} ${4:-1024} >> $out 2>&1
  for v in zstd_amd/variants/*.so; do
    [ -f "$v" ] && ZHIP_LIB=$PWD/$v timeout 600 python scripts/ab_parse.py ${2:-1} ${3:-silesia,text,datagen} ${4:-1024} >> $out 2>&1
  done
done
grep '^{' $out | python3 -c "
import sys, json, collections
best = collections.OrderedDict()
for l in sys.stdin:
    d = json.loads(l); k = (d['lib'], d['shape'])
    if k not in best or d['parse_ms'] < best[k]['parse_ms']: best[k] = d
for (lib, shape), d in best.items(): print('%-28s %-8s parse %8.3f ms  entropy %6.3f  %6.2f GB/s  sha %s' % (lib, shape, d['parse_ms'], d['entropy_ms'], d['GBps'], d['sha']))
"
