#!/bin/bash
# round 6: frame kernels A/B: scripts/frames_timing.py on the product and on every zstd_amd/variants/*.so
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/ab_frames_${1:-x}.log
: > $out
for v in "" zstd_amd/variants/*.so; do
  [ -n "$v" ] && [ ! -f "$v" ] && continue
  echo "## ${v:-libzstd_hip.so}" >> $out
  ZHIP_LIB=${v:+$PWD/$v} timeout 600 python scripts/frames_timing.py 2>&1 | grep '^{' >> $out
done
python3 - <<PY
import json
lib=None
for l in open("$out"):
    if l.startswith("##"): lib=l[3:].strip(); continue
    d=json.loads(l); t=d["timing_ms"]
    print("%-40s %-8s %5d x %8d  %s" % (lib, d["kind"], d["frames"], d["frame_bytes"], {k: round(v,2) for k,v in t.items()}))
PY
