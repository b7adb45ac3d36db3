#!/usr/bin/env bash
# A/B of the register budget of the wide counter classes: 4 waves/SIMD with 5-14 spilled VGPRs vs 3
# waves without scratch for the 16-plane class (queries of 8 192 ... 65 535 k-mers: 9.7 kbp and 30 kbp contigs), and 3 vs 2
# waves for the wide-query form of the 20-plane class (100 kbp).  12 M k-mers per step, config 3, both scan modes.
#   GPU box: bash tools/ab_wide_counter_waves.sh > gpurun_out/r05/ab_wide_counter_waves.txt
run() {  # flags label
  PM_EXTRA_FLAGS="$1" python3 phylign_amd/build.py > /dev/null 2>&1 || { echo "build failed: $1"; return; }
  for a in "--queries 1240 --qlen 9700" "--queries 400 --qlen 30030" "--queries 120 --qlen 100030"; do
    for mode in fetch_all_rows threshold_bound; do
      python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --whole-record --only-headline --headline $mode $a 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%s] %s %s: %.1f Mkmers/s %.2f ms' % (sys.argv[1], sys.argv[2], sys.argv[3], d['value']/1e6, d['ms_per_step']), {k: round(v['avg_ms'],2) for k,v in d['scan_launches'].items()})" "$1" "$a" "$mode"
    done
  done
}
run ""
run "-DPM_SCAN_WAVES_P16=3"
run "-DPM_SCAN_WAVES_P20_WQ=2"
PM_EXTRA_FLAGS="" python3 phylign_amd/build.py > /dev/null 2>&1
