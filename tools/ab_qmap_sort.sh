#!/usr/bin/env bash
# A/B: queries of a counter class in file order (PM_QMAP_SORT=0) vs longest first (default), on the gene-length leg
# (the length mix of data/ARGannot_r3.fa, x1 and x8).  GPU box: bash tools/ab_qmap_sort.sh > gpurun_out/r05/ab_qmap_sort.txt
for rep in 1 2; do
for s in 0 1; do
  PM_QMAP_SORT=$s python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-live-pmc --whole-record --no-full-shard --no-clustered --no-l31 --no-unique-rows 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for x in ('x1','x8'):
    g=d['argannot'][x]
    for m in ('fetch_all_rows','threshold_bound'):
        print('PM_QMAP_SORT=%s argannot %s %s: %.3f ms/step %.1f Mkmers/s' % (sys.argv[1], x, m, g[m]['ms_per_step'], g[m]['value']/1e6), {k: round(v['avg_ms'],3) for k,v in g[m]['scan_launches'].items()})" $s
done
done
