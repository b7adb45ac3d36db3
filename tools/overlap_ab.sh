#!/usr/bin/env bash
# A/B of pm_set_option("overlap_launches"): the mixed-width launch on a second stream beside the wide ones.
# N = 1 headline and ranks of an emulated 8-way split, both scan modes.   bash tools/overlap_ab.sh > gpurun_out/overlap_ab.txt
for ov in 0 1; do
  for mode in fetch_all_rows threshold_bound; do
    for spec in "1 0" "8 0" "8 1" "8 2" "8 5"; do
      set -- $spec
      extra=""; [ "$1" != 1 ] && extra="--emulate-world $1 --emulate-rank $2"
      PM_OVERLAP_LAUNCHES=$ov python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --only-headline $extra --headline $mode 2>/dev/null \
        | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('overlap=$ov $mode world=$1 rank=$2 ms_per_step=%.3f kernels=%.3f' % (d['ms_per_step'], d['rank0_ms']['kernels_total']))"
    done
  done
done
