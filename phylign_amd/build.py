"""Builds libphylign_match.so (HIP, gfx950 only) in-tree with hipcc.

The library is the product's only compute path; there is no fallback build."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libphylign_match.so")
# measurement / test aids (synthetic indexes, planted hits, gather probe): a library of its own that reaches the product
# only through its C ABI -- loaded by bench.py, tools/ and tests/, never by the drop-in path
AIDS_LIB = os.path.join(_HERE, "libphylign_bench.so")
AIDS_SOURCES = [os.path.join("bench", "pm_bench_aids.hip")]
AIDS_HEADERS = [os.path.join("..", "..", "include", "phylign_match_bench.h")]
SOURCES = ["pm_kernels.hip", "pm_runtime.cpp", "pm_index.cpp", "pm_queries.cpp", "pm_search.cpp", "pm_text.cpp", "pm_gzfast.cpp"]
HEADERS = ["pm_internal.h", "pm_host.h", "exports.map", os.path.join("..", "..", "include", "phylign_match.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; this package builds for gfx950 only)")


def is_stale():
    if not os.path.exists(LIB) or not os.path.exists(AIDS_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(AIDS_LIB))
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS + AIDS_SOURCES + AIDS_HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the HIP kernels + host C++ into phylign_amd/libphylign_match.so."""
    if not force and not is_stale():
        return LIB
    def compile_one(src):
        o = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
               "-x", "hip", "-c", os.path.join(CSRC, src), "-o", o] + os.environ.get("PM_EXTRA_FLAGS", "").split()
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES) + len(AIDS_SOURCES), os.cpu_count() or 1)) as ex:
        all_objs = list(ex.map(compile_one, SOURCES + AIDS_SOURCES))
    objs, aid_objs = all_objs[:len(SOURCES)], all_objs[len(SOURCES):]
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lz", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    # the aids library links against the product library next to it (rpath $ORIGIN) and uses only its exported C ABI
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", AIDS_LIB] + aid_objs + \
          ["-L" + _HERE, "-lphylign_match", "-Wl,-rpath,$ORIGIN", "-Wl,--version-script=" + os.path.join(CSRC, "bench", "exports.map")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
