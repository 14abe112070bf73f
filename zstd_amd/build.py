"""Builds zstd_amd/libzstd_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and the tests.

One object per translation unit — zhip_lib.hip (host side: the C ABI, launches) and one zhip_k_<family>.hip per kernel family — compiled in
parallel and linked into one shared library.  Each family is its own code object, so a change in one cannot move another's code."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libzstd_hip.so")
SHIM = os.path.join(HERE, "libzstd_hipshim.so")      # ZSTD_*-named drop-in (plain C) on top of LIB
WORKLOADS = os.path.join(HERE, "libzhip_workloads.so")   # bench / test input generators (host C++, NOT part of the product library)
UNITS = ["zhip_lib", "zhip_k_parse", "zhip_k_lazy", "zhip_k_entropy", "zhip_k_frames", "zhip_k_decode"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]
# per-unit code generation options, each an A/B on MI355X (profiles/r04_ab_sched_strategies.log, r04_ab_maxilp.log): the machine scheduler's max-ILP strategy
# shortens the match finders' dependent chains (ZSTD_fast stage: datagen -3 %, text -2 %, Silesia-shaped -8 %; frame kernels -3 %; decoder -1.5 %); the entropy
# stage and ZSTD_dfast (bound by table-line latency) do not move, the lazy unit kernels were not measured and keep the default
UNIT_FLAGS = {u: ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] for u in ("zhip_k_parse", "zhip_k_frames", "zhip_k_decode")}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def sources():
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip", ".cpp"))]        # zstd_shim.c is the shim's only source
    deps.append(os.path.join(HERE, "..", "include", "zstd_hip.h"))
    return deps


def hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_unit(name, extra=(), verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    obj = os.path.join(OBJ, name + ".o")
    cmd = [hipcc()] + FLAGS + UNIT_FLAGS.get(name, []) + list(extra) + ["-c", os.path.join(CSRC, name + ".hip"), "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    """compile every HIP source for gfx950 into one shared library; returns its path"""
    if force or _stale(LIB, sources()):
        with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 2)) as ex:
            objs = list(ex.map(lambda u: compile_unit(u, verbose=verbose), UNITS))
        cmd = [hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    shim_src = os.path.join(CSRC, "zstd_shim.c")
    if force or _stale(SHIM, [shim_src, LIB, os.path.join(HERE, "..", "include", "zstd_hip_dropin.h")]):
        cmd = [os.environ.get("CC", "gcc"), "-O2", "-std=c99", "-Wall", "-Wextra", "-fPIC", "-shared", shim_src, "-o", SHIM,
               "-L" + HERE, "-lzstd_hip", "-Wl,-rpath,$ORIGIN"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    wl_dir = os.path.join(HERE, "workloads_src")
    wl_src = [os.path.join(wl_dir, f) for f in sorted(os.listdir(wl_dir))]
    if force or _stale(WORKLOADS, wl_src):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-fPIC", "-shared", os.path.join(wl_dir, "workloads_lib.cpp"), "-o", WORKLOADS]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
