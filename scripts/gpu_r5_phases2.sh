#!/bin/bash
# round 5: s_memtime phase split (measurement build -DZHIP_PROF) of the text level-1 parse and of the level-5 parse, one GPU call
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
rm -f gpurun_out/r05/phases2.log
for cfg in "text 1 1024" "datagen 5 1024" "text 5 1024"; do
  set -- $cfg
  echo "== WORKLOAD=$1 LEVEL=$2 MIB=$3" >> gpurun_out/r05/phases2.log
  WORKLOAD=$1 LEVEL=$2 MIB=$3 timeout 300 python scripts/prof_phases.py 2>/dev/null >> gpurun_out/r05/phases2.log
done
cat gpurun_out/r05/phases2.log
