"""first-contact probe on the GPU box: environment facts + parse-kernel timing (not the bench contract)."""
import os, sys, time, json, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
import zstd_amd
from _libs import load_oracle, datagen

print("ref on box:", os.path.exists("/root/reference/lib/zstd.h"), "| nproc", os.cpu_count(), "| torch", torch.__version__)
p = torch.cuda.get_device_properties(0)
print("gpu:", p.name, "CUs", p.multi_processor_count, "mem GB", p.total_memory / 2**30)
lo = load_oracle()
units = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n = units * 131072
t = time.time(); a = datagen(lo, n, P, 0); print("datagen s", time.time() - t)
d = torch.from_numpy(np.concatenate([a, np.zeros(64, np.uint8)])).cuda()
ctx = zstd_amd.Context(0, max_units=units)
for it in range(4):
    ctx.parse_device(d.data_ptr(), n, 1, 131072)
    tm = ctx.timing()
    print(f"parse {units} units P{P}: {tm['parse_ms']:.3f} ms -> {n / tm['parse_ms'] / 1e6:.2f} GB/s")
