#!/usr/bin/env bash
# A/B of the one-step-ahead hash load of the sharing k_scan instantiations (PM_SCAN_PREFETCH_HASH): off, on for every
# sharing class (<= 13 planes; the 13-plane class then keeps 1-5 spilled VGPRs), on for <= 10 planes only.  Headline
# (7 planes) and the gene-length leg (10 / 13 planes), both scan modes.
#   GPU box: bash tools/ab_prefetch_hash.sh > gpurun_out/r06/ab_prefetch_hash.txt
for rep in 1 2; do
for v in "-DPM_SCAN_PREFETCH_HASH=0" "-DPM_SCAN_PREFETCH_HASH=1 -DPM_SCAN_PREFETCH_MAX_P=13" "-DPM_SCAN_PREFETCH_HASH=1 -DPM_SCAN_PREFETCH_MAX_P=10"; do
  PM_EXTRA_FLAGS="$v" python3 phylign_amd/build.py > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc --whole-record --no-full-shard --no-clustered --no-l31 --no-unique-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
tag='[%s]' % sys.argv[1]
print(tag, 'headline fetch_all: %.3f ms/step' % d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['scan_launches'].items()})
b=d['threshold_bound']
print(tag, 'headline bound:     %.3f ms/step' % b['ms_per_step'])
for x in ('x1','x8'):
    g=d['argannot'][x]
    for m in ('fetch_all_rows','threshold_bound'):
        print(tag, 'argannot %s %s: %.3f ms/step %.1f Mkmers/s' % (x, m, g[m]['ms_per_step'], g[m]['value']/1e6), {k: (round(v['avg_ms'],3), round(v['algorithmic_GBps'])) for k,v in g[m]['scan_launches'].items()})" "$v"
done
done
PM_EXTRA_FLAGS="" python3 phylign_amd/build.py > /dev/null 2>&1
