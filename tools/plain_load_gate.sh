#!/usr/bin/env bash
# How many plain index files should stream into HBM at once?  One rank's full-size shard (38 files, 128 GB) is written to
# a memory-backed directory once, then `match_stage` runs on it with PHYLIGN_PLAIN_LOADS = 12 1 2 3 4 6 (12 = every loader
# thread at once, the behaviour before round 5), twice.  GPU box: bash tools/plain_load_gate.sh [rows-divisor] > gpurun_out/r05/plain_load_gate.txt
# (profiles/r05/plain_load_gate_sweep_2_defer_free.txt was made by a variant of this script that also set PHYLIGN_DEFER_FREE,
#  a stage option that was measured and not kept.)
div=${1:-1}
work=/dev/shm/plg
python3 tools/e2e_cold_warm.py --rows-divisor "$div" --modes cached --work $work --queries 100000 --keep > /dev/null 2>&1 || { echo "setup failed"; rm -rf $work; exit 1; }
for rep in 1 2; do
for g in 12 1 2 3 4 6; do
  rm -rf $work/03_x $work/04_x
  PHYLIGN_PLAIN_LOADS=$g PYTHONPATH=$PWD python3 -m phylign_amd.match_stage --batches $work/batches.txt --cobs-dir $work/cobs --sizes $work/sizes.txt \
      --queries $work/Q.fa --out-dir $work/03_x --filter-out $work/04_x/Q.fa --cache-dir $work/cache 2>&1 >/dev/null | python3 -c "
import json,sys
rep=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('plain loads at once %s: e2e %.2f s, stage %.2f s, load thread-sum %.1f s, groups %d' % (sys.argv[1], rep['e2e_s'], rep['stage_wall_s'], rep['load_s_thread_sum'], rep['groups']))" $g
done
done
rm -rf $work
