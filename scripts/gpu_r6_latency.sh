#!/bin/bash
# round 6: how long ONE unit takes by the number of units resident beside it — text (the heaviest shape) at 128 / 256 / 512 / 1024 MiB = 1 024 ... 8 192 units on 4 096 slots
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/unit_latency.log
: > $out
for mib in 64 128 256 384 512 768 1024; do
  timeout 300 python scripts/ab_parse.py 1 text,silesia,datagen $mib >> $out 2>&1
done
grep '^{' $out | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-8s %5d MiB  parse %8.3f ms' % (d['shape'], d['MiB'], d['parse_ms']))
"
