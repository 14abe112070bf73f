#!/bin/bash
# round 6: host-buffer path sweep (scripts/e2e_sweep.py) for the product library and every zstd_amd/variants/e2e_*.so
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/e2e_${1:-x}.log
: > $out
for lib in zstd_amd/libzstd_hip.so zstd_amd/variants/e2e_*.so; do
  [ -f "$lib" ] || continue
  echo "## $lib" >> $out
  ZHIP_LIB=$PWD/$lib LANES=${2:-2} CUS=${3:-1024,1536,2048} timeout 600 python scripts/e2e_sweep.py >> $out 2>&1
done
grep -v amdgpu.ids $out | cut -c1-160
