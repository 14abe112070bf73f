#!/bin/bash
# scripts/build_variant.sh NAME UNIT [hipcc flags...] — measurement helper: another build of the product library into zstd_amd/variants/NAME.so in which the
# translation unit UNIT (zhip_k_parse, zhip_k_entropy, ... or zhip_lib) is recompiled with the given flags; the other objects are the product's (zstd_amd/build/)
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
mkdir -p zstd_amd/variants /tmp/zhip_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result "$@" -c zstd_amd/csrc/$unit.hip -o /tmp/zhip_variants/$name.o
objs=""; for u in zhip_lib zhip_k_parse zhip_k_lazy zhip_k_entropy zhip_k_frames zhip_k_decode; do if [ $u = $unit ]; then objs="$objs /tmp/zhip_variants/$name.o"; else objs="$objs zstd_amd/build/$u.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared $objs -o zstd_amd/variants/$name.so
echo built zstd_amd/variants/$name.so
