#!/bin/bash
# scripts/build_variant.sh NAME [hipcc flags...] — measurement helper: another build of the product library into zstd_amd/variants/NAME.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p zstd_amd/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function -Wno-unused-result "$@" zstd_amd/csrc/zhip_lib.hip -o zstd_amd/variants/$name.so
echo built zstd_amd/variants/$name.so
