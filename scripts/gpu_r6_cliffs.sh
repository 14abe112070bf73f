#!/bin/bash
# round 6: every member shape (scripts/l5_components.py, EXTRA=1) at several levels through the unit path: a shape that takes many times the others' time is a cliff
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r06
out=gpurun_out/r06/cliffs.log
: > $out
for lvl in ${@:-1 3 5 7 10 -1}; do
  EXTRA=1 timeout 900 python scripts/l5_components.py $lvl 64 2>&1 | grep '^{' >> $out
done
python3 - <<PY
import json, collections
t = collections.OrderedDict()
for l in open("$out"):
    d = json.loads(l); t.setdefault(d["member"], {})[d["level"]] = d["parse_ms_per_GiB"]
lv = sorted({k for v in t.values() for k in v})
print("%-16s" % "ms per GiB" + "".join("%10s" % ("L%d" % k) for k in lv))
for m, v in t.items(): print("%-16s" % m + "".join("%10.0f" % v.get(k, float("nan")) for k in lv))
PY
