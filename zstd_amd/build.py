"""Builds zstd_amd/libzstd_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and the tests."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libzstd_hip.so")
SHIM = os.path.join(HERE, "libzstd_hipshim.so")      # ZSTD_*-named drop-in (plain C) on top of LIB


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def sources():
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip", ".cpp"))]        # zstd_shim.c is the shim's only source
    deps.append(os.path.join(HERE, "..", "include", "zstd_hip.h"))
    return deps


def build(force=False, verbose=False):
    """compile every HIP source for gfx950 into one shared library; returns its path"""
    if force or _stale(LIB, sources()):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
               "-Wno-unused-function", "-Wno-unused-result",
               os.path.join(CSRC, "zhip_lib.hip"), "-o", LIB]
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
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
