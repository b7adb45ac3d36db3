#!/usr/bin/env bash
# A/B of the row-index sharing (k_scan SHARE) by counter class: PM_SCAN_SHARE_MAX_P = 13 (round 6 default: the 7-, 10- and
# 13-plane classes share 32-bit row indices), 7 (round 5's reach, with the 32-bit exchange) and 0 (no sharing), on the
# headline (7 planes) and on the gene-length leg (10 / 13 planes; x8 = 12.8 M k-mers).
#   GPU box: bash tools/ab_share_rows.sh > gpurun_out/r06/ab_share_rows.txt
for rep in 1 2; do
for maxp in 13 7 0; do
  PM_EXTRA_FLAGS="-DPM_SCAN_SHARE_MAX_P=$maxp" python3 phylign_amd/build.py > /dev/null 2>&1 || { echo "build failed: $maxp"; continue; }
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc --whole-record --no-full-shard --no-clustered --no-l31 --no-unique-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
tag='SHARE_MAX_P=%s' % sys.argv[1]
print(tag, 'headline fetch_all: %.3f ms/step' % d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['scan_launches'].items()})
b=d['threshold_bound']
print(tag, 'headline bound:     %.3f ms/step' % b['ms_per_step'])
for x in ('x1','x8'):
    g=d['argannot'][x]
    for m in ('fetch_all_rows','threshold_bound'):
        print(tag, 'argannot %s %s: %.3f ms/step %.1f Mkmers/s' % (x, m, g[m]['ms_per_step'], g[m]['value']/1e6), {k: (round(v['avg_ms'],3), round(v['algorithmic_GBps'])) for k,v in g[m]['scan_launches'].items()})" $maxp
done
done
PM_EXTRA_FLAGS="" python3 phylign_amd/build.py > /dev/null 2>&1
