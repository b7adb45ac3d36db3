#!/usr/bin/env bash
# A/B of build-time kernel variants on the GPU box: tools/variant_bench.sh "<flags A>" "<flags B>" ...
# rebuilds libphylign_match.so with each PM_EXTRA_FLAGS set and prints step / dominant-kernel times.
for flags in "$@"; do
  PM_EXTRA_FLAGS="$flags" python3 phylign_amd/build.py > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-pmc --whole-record --no-clustered 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['threshold_bound']
print('flags=[%s] fetch-all: step %.2f ms, k_scan<32> %.2f ms | bound: step %.2f ms, k_scan<32> %.2f ms' % (sys.argv[1], d['ms_per_step'], d['roofline']['avg_launch_ms'], b['ms_per_step'], b['roofline']['avg_launch_ms']))" "$flags"
done
