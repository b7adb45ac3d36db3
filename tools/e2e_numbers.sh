#!/usr/bin/env bash
# The config-4/5 end-to-end rows of profiles/r03/NOTES.md section 6 (GPU box, repo root):  bash tools/e2e_numbers.sh <outdir>
set -u
out=${1:-gpurun_out/e2e_now}
mkdir -p $out
E2E="python3 tools/e2e_config5.py --out gpurun_out/e2e_x"
$E2E --queries 100000 --max-group 0 1 > $out/e2e_config4_100k.json 2>/dev/null
$E2E --queries 1000000 --max-group 0 38 1 > $out/e2e_config5_1M_iid.json 2>/dev/null
$E2E --queries 1000000 --clustered --max-group 0 38 1 > $out/e2e_config5_1M_clustered.json 2>/dev/null
$E2E --queries 1000000 --bound 0 --max-group 0 > $out/e2e_config5_1M_iid_fetch_all.json 2>/dev/null
$E2E --queries 4000000 --query-chunk 1000000 --max-group 0 > $out/e2e_config5_4M_in_4_chunks.json 2>/dev/null
rm -rf gpurun_out/e2e_x
for f in $out/e2e*.json; do echo $f; python3 -c "
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print({k:d.get(k) for k in ('max_group','query_chunks','groups','scan_launches','parse_queries_s','match_only_s','d2h_s','format_s_thread_sum','merge_s_thread_sum','stage_wall_s','filter_emit_s','e2e_s','gz_bytes','filter_fasta_bytes')}, d['config'][:60])
" $f; done
