#!/bin/bash
# round 5: s_memtime phase split of the dfast and fast parsers (measurement build -DZHIP_PROF), one GPU call
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05
for cfg in "silesia 3 4096" "datagen 3 1024" "text 3 1024" "silesia 1 2048" "datagen 1 1024"; do
  set -- $cfg
  echo "== WORKLOAD=$1 LEVEL=$2 MIB=$3" >> gpurun_out/r05/phases.log
  WORKLOAD=$1 LEVEL=$2 MIB=$3 timeout 300 python scripts/prof_phases.py >> gpurun_out/r05/phases.log 2>&1
done
tail -c 3000 gpurun_out/r05/phases.log
