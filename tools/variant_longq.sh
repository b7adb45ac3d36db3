for v in "-DPM_SCAN_WAVES_P16=4 -DPM_SCAN_WAVES_P24=4" "-DPM_SCAN_WAVES_P16=3 -DPM_SCAN_WAVES_P24=3" "-DPM_SCAN_WAVES_P16=4 -DPM_SCAN_WAVES_P24=2"; do
  PM_EXTRA_FLAGS="$v" python3 -m phylign_amd.build > /dev/null 2>&1
  for a in "--queries 1240 --qlen 9700" "--queries 120 --qlen 100030"; do
    python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --only-headline $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"$v | $a\", round(d[\"value\"]/1e6,1), round(d[\"ms_per_step\"],2), {k:round(v[\"avg_ms\"],2) for k,v in d[\"scan_launches\"].items()})"
  done
done
