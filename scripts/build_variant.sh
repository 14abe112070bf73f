#!/bin/bash
# scripts/build_variant.sh NAME UNIT[,UNIT...] [hipcc flags...] — measurement helper: another build of the product library into zstd_amd/variants/NAME.so in which the
# translation units UNIT (zhip_k_parse, zhip_k_entropy, ... or zhip_lib) are recompiled with the given flags ON TOP of the unit's own product options
# (zstd_amd/build.py UNIT_FLAGS); the other objects are the product's (zstd_amd/build/).  A variant that changes a constant the host side also reads
# (LDS sizes) names both units: zhip_k_lazy,zhip_lib
set -e
cd "$(dirname "$0")/.."
name=$1; units=$2; shift 2
mkdir -p zstd_amd/variants /tmp/zhip_variants
objs=""
for u in zhip_lib zhip_k_parse zhip_k_lazy zhip_k_entropy zhip_k_frames zhip_k_decode; do
  if [[ ",$units," == *",$u,"* ]]; then
    uf=$(python3 -c "import sys; sys.path.insert(0,'.'); from zstd_amd.build import UNIT_FLAGS; print(' '.join(UNIT_FLAGS.get('$u', [])))")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result $uf "$@" -c zstd_amd/csrc/$u.hip -o /tmp/zhip_variants/${name}_$u.o &
    objs="$objs /tmp/zhip_variants/${name}_$u.o"
  else
    objs="$objs zstd_amd/build/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs -o zstd_amd/variants/$name.so
echo built zstd_amd/variants/$name.so
