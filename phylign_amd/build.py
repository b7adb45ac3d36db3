"""Builds libphylign_match.so (HIP, gfx950 only) in-tree with hipcc.

The library is the product's only compute path; there is no fallback build."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libphylign_match.so")
SOURCES = ["pm_kernels.hip", "pm_runtime.cpp", "pm_index.cpp", "pm_queries.cpp", "pm_search.cpp", "pm_text.cpp", "pm_gzfast.cpp"]
HEADERS = ["pm_internal.h", "pm_host.h", os.path.join("..", "..", "include", "phylign_match.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; this package builds for gfx950 only)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the HIP kernels + host C++ into phylign_amd/libphylign_match.so."""
    if not force and not is_stale():
        return LIB
    def compile_one(src):
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
               "-x", "hip", "-c", os.path.join(CSRC, src), "-o", o] + os.environ.get("PM_EXTRA_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
