#!/usr/bin/env python3
"""scripts/scan_divergent_barriers.py — static check of the gfx950 ISA: loops whose back edge is taken on EXEC (s_cbranch_execnz), i.e.
loops the compiler treats as lane-divergent, that contain s_barrier.  Such a loop is only safe while every lane of a wavefront leaves
it in the same iteration (a uniform value the compiler could not prove uniform).  The prefix fill of the 24-bit frame table hung on the
GPU exactly there: the compiler split a "retry until nobody lost" loop per lane and the flag-resetting lane sat masked
while its wave went on to the next barrier.  Usage: python scripts/scan_divergent_barriers.py  (compiles zhip_lib.hip to assembly)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "zhip_lib_scan.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                       os.path.join(ROOT, "zstd_amd", "csrc", "zhip_lib.hip"), "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
labels, funcs = {}, []
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
    m = re.match(r"^(_ZN4zhip\d+)(\w+?)E", l)
    if m and l.rstrip().endswith(":") or (m and ": " in l):
        funcs.append((i, m.group(2)))


def fn(i):
    name = "?"
    for j, f in funcs:
        if j <= i:
            name = f
    return name


found = 0
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch_execnz\s+(\.LBB\d+_\d+)", l)
    if not m:
        continue
    t = labels.get(m.group(1))
    if t is None or t >= i or "Loop Header" not in " ".join(lines[t:t + 2]):       # LLVM annotates loop header blocks; other backward targets are layout
        continue
    nb = sum(1 for x in lines[t:i] if "s_barrier" in x)
    if nb:
        found += 1
        print(f"{fn(i):24s} loop of {i - t:6d} instructions lines, {nb:2d} s_barrier inside")
print("EXEC-controlled loops containing s_barrier:", found, "(each must exit all lanes of a wavefront together)")

# second check (round 3): k_decode must stay ONE piece of code.  When the block-parallel decoder became a second caller of dec_huf_table /
# dec_huf_streams_par the compiler stopped inlining them; every GPU step that then used k_decode did not come back ("the last GPU call").
body = []
inside = False
for l in lines:
    if re.match(r"^_ZN4zhip8k_decodeE\w+:", l):
        inside = True
    elif inside and l.startswith(".Lfunc_end"):
        break
    if inside:
        body.append(l)
calls = sum(1 for l in body if "s_swappc_b64" in l)
print("k_decode function calls:", calls, "(must be 0: its callees are always_inline)")
