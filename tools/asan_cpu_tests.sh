#!/usr/bin/env bash
# Host code of libphylign_match.so under AddressSanitizer + UBSan, against the CPU test suite
# (parsers, header readers, text formatting, 04_filter merge, corruption fuzz).  GPU sanitizers are not
# available on this pool, so this covers the host translation units only; the kernels are built as usual.
#   bash tools/asan_cpu_tests.sh        (from the repo root; needs no GPU)
#   bash tools/asan_cpu_tests.sh thread (ThreadSanitizer instead: the worker pool, the parallel parser / formatter, the merge)
set -euo pipefail
kind=${1:-address}
if [ "$kind" = thread ]; then san=thread; rt=tsan; else san=address,undefined; rt=asan; fi
out=${TMPDIR:-/tmp}/pm_$rt
mkdir -p "$out"
cd "$(dirname "$0")/../phylign_amd/csrc"
for f in pm_runtime pm_index pm_queries pm_search pm_text pm_gzfast; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=$san -fno-omit-frame-pointer \
      -Wno-option-ignored -x hip -c $f.cpp -o "$out/$f.o" 2>/dev/null &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c pm_kernels.hip -o "$out/pm_kernels.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=$san -o "$out/libphylign_match.so" "$out"/*.o -lz -Wl,--version-script=exports.map
cd ../..
asan=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.$rt-x86_64.so" | head -1)
PHYLIGN_MATCH_LIB="$out/libphylign_match.so" LD_PRELOAD="$asan" ASAN_OPTIONS=detect_leaks=0 TSAN_OPTIONS=report_signal_unsafe=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest tests/test_golden_cpu.py -q -p no:cacheprovider
