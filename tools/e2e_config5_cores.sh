#!/usr/bin/env bash
# Config 5 (1 M queries on one rank's resident 135 GB shard -> 38 .gz + the 04_filter FASTA) with the host share a rank
# of an 8-rank node gets: tools/e2e_config5.py as a FRESH process under `taskset` to 2 / 4 / 8 / 16 CPUs, with and
# without the per-batch files (--filter-only).  GPU box: bash tools/e2e_config5_cores.sh gpurun_out/r05/e2e_config5_cores.jsonl
out=${1:-gpurun_out/r05/e2e_config5_cores.jsonl}
mkdir -p "$(dirname "$out")"; : > "$out"
cpus=$(python3 -c "import os; print(','.join(str(c) for c in sorted(os.sched_getaffinity(0))))")
for n in 2 4 8 16; do
  set_=$(python3 -c "import sys; c=sys.argv[1].split(','); print(','.join(c[:int(sys.argv[2])]))" "$cpus" "$n")
  for extra in "" "--filter-only"; do
    taskset -c "$set_" python3 tools/e2e_config5.py --queries 1000000 --piece-mb 48 --out /dev/shm/e2e_c5 --json "$out" $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cpus', d['host_cpus'], 'filter_only', d['filter_only'], 'e2e %.3f s  match_only %.3f s  stage %.3f s  emit %.3f s' % (d['e2e_s'], d['match_only_s'], d['stage_wall_s'], d['filter_emit_s']))"
    rm -rf /dev/shm/e2e_c5
  done
done
