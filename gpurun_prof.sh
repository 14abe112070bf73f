cd /root/repo
for W in datagen text silesia; do echo "== $W"; WORKLOAD=$W MIB=512 timeout 200 python scripts/prof_phases.py 2>&1 | grep -v amdgpu.ids; done
