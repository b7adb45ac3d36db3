#!/usr/bin/env bash
# Per-rank step time of an N-way strong-scaling split, one shard at a time on ONE GPU
# (bench.py --emulate-world N --emulate-rank r): the slowest rank bounds the N-GPU step
# before the RCCL gather.
# Usage (GPU box): bash tools/emulate_scaling.sh ["1 2 4 8" [extra bench.py flags ...]] > gpurun_out/emulate_scaling.tsv
worlds=${1:-"1 2 4 8"}
shift || true
echo -e "mode\tworld\trank\tbatches\tms_per_step\tkernels_ms\tshared"
for mode in fetch_all_rows threshold_bound; do
  for n in $worlds; do
    for ((r=0; r<n; r++)); do
      python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --whole-record --emulate-world $n --emulate-rank $r --headline $mode "$@" 2>/dev/null \
        | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode\t$n\t$r\t%d\t%.3f\t%.3f\t%d' % (sum(v['batches'] for v in d['scan_launches'].values()), d['ms_per_step'], d['rank0_ms']['kernels_total'], len(d['config']['batches_on_two_ranks'])))"
    done
  done
done
