#!/usr/bin/env bash
# Longer queries at the same 12 M k-mers per step: k-mers/s of the fetch-all and the bounded scan, with the
# wide-query form automatic (0), forced (1) and off (2).  GPU box: bash tools/variant_longq.sh
for mode in fetch_all_rows threshold_bound; do
for wq in 0 1 2; do
  for a in "--queries 100000 --qlen 150" "--queries 12400 --qlen 1000" "--queries 4000 --qlen 3100" "--queries 1240 --qlen 9700" "--queries 120 --qlen 100030"; do
    PM_WIDE_QUERY=$wq python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --whole-record --only-headline --headline $mode $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$mode wq=$wq $a\", round(d[\"value\"]/1e6,1), \"Mkmers/s\", round(d[\"ms_per_step\"],2), \"ms\", {k:round(v[\"avg_ms\"],2) for k,v in d[\"scan_launches\"].items()})"
  done
done
done
