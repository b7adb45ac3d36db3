#!/usr/bin/env bash
# Where does a plain-file load spend its time?  One rank's shard at rows / $1 as files in /dev/shm, the stage run a few times
# with PM_LOAD_TRACE=1 (per load: hipMalloc ms; pread vs staging-wait per reader), plus numactl-style placement facts.
div=${1:-2}
work=/dev/shm/plt
python3 tools/e2e_cold_warm.py --rows-divisor "$div" --modes cached --work $work --queries 100000 --keep > /dev/null 2>&1 || { echo "setup failed"; rm -rf $work; exit 1; }
echo "GPU numa node: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"
for f in /sys/devices/system/node/node*/cpulist; do echo "$f: $(cat $f)"; done
grep -E "^Node [01] (FilePages|Shmem|MemFree):" /sys/devices/system/node/node*/meminfo 2>/dev/null | head -8
for run in 1 2 3 4; do
  rm -rf $work/03_x $work/04_x
  PM_LOAD_TRACE=1 PHYLIGN_PLAIN_LOADS=${GATE:-2} PYTHONPATH=$PWD python3 -m phylign_amd.match_stage --batches $work/batches.txt --cobs-dir $work/cobs --sizes $work/sizes.txt \
      --queries $work/Q.fa --out-dir $work/03_x --filter-out $work/04_x/Q.fa --cache-dir $work/cache > /dev/null 2> $work/err.txt
  python3 - $work/err.txt $run <<'PY'
import json, re, sys
txt = open(sys.argv[1]).read()
rep = json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
mal = [float(m.group(2)) for m in re.finditer(r"hipMalloc ([\d.]+) GB: ([\d.]+) ms", txt)]
loads = [(float(m.group(1)), float(m.group(2)), float(m.group(3)), float(m.group(5)), float(m.group(6))) for m in
         re.finditer(r"\] ([\d.]+) GB in ([\d.]+) s = ([\d.]+) GB/s; (\d+) readers: pread ([\d.]+) s, waiting for a staging buffer ([\d.]+) s", txt)]
print("run %s: e2e %.2f s; hipMalloc total %.0f ms (max %.0f); %d traced loads: sum wall %.2f s, mean %.1f GB/s, pread share %.2f, staging-wait share %.2f" % (
    sys.argv[2], rep["e2e_s"], sum(mal), max(mal or [0]), len(loads), sum(l[1] for l in loads), sum(l[0] for l in loads) / max(sum(l[1] for l in loads), 1e-9),
    sum(l[3] for l in loads) / max(sum(l[1] for l in loads), 1e-9), sum(l[4] for l in loads) / max(sum(l[1] for l in loads), 1e-9)))
PY
done
rm -rf $work
