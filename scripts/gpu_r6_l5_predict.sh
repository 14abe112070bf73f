#!/bin/bash
# round 6: level 5 on the three shapes with the units' two-pass prediction off / on, and where the Silesia-shaped mix's time goes by component
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/l5_predict.log
: > $out
for p in 0 1; do
  echo "## PREDICT=$p" >> $out
  PREDICT=$p timeout 900 python scripts/ab_parse.py 5 silesia,text,datagen 1024 2>&1 | grep '^{' >> $out
done
cat $out | cut -c1-330
