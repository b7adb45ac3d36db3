#!/usr/bin/env bash
# Few, very long queries at 12 M k-mers per step (fetch-all scan): the wide-query form with the steps of a query
# shared by workgroups (automatic) and not (PM_WQ_SPLIT=1).  GPU box: bash tools/longq_split.sh
for split in 1 0; do
  for a in "--queries 1240 --qlen 9700" "--queries 120 --qlen 100030" "--queries 40 --qlen 300030" "--queries 12 --qlen 1000030" "--queries 3 --qlen 4000030"; do
    PM_WQ_SPLIT=$split python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --whole-record --only-headline $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"split=$split $a\", round(d[\"value\"]/1e6,1), \"Mkmers/s\", round(d[\"ms_per_step\"],2), \"ms\", {k:round(v[\"avg_ms\"],2) for k,v in d[\"scan_launches\"].items()})"
  done
done
