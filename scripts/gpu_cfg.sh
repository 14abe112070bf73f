#!/bin/bash
# scripts/gpu_cfg.sh — GPU box: the BASELINE.json configs that fit one GPU, on their synthetic stand-ins
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
run() { name=$1; shift; timeout 600 python bench.py "$@" > gpurun_out/cfg_$name.json 2> gpurun_out/cfg_$name.err; echo "$name rc=$?"; tail -c 2500 gpurun_out/cfg_$name.json; grep -v amdgpu.ids gpurun_out/cfg_$name.err | tail -3; }
run c3_silesia64_L3 --workload silesia --copies 64 --level 3 --steps 2 --warmup 1
run c3_silesia8_L1 --workload silesia --copies 8 --level 1 --steps 3 --warmup 1
run c4_text_L1 --workload text --total-bytes 1000000000 --level 1 --steps 3 --warmup 1
