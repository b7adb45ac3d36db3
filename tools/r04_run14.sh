mkdir -p gpurun_out/r04
echo -e "mode\tworld\trank\tbatches\tms_per_step\tkernels_ms\tkmers_per_s" > gpurun_out/r04/emulate_full_collection_n8.tsv
for mode in fetch_all_rows threshold_bound; do
for r in 0 1 2 3 4 5 6 7; do
python3 bench.py --workload full --steps 10 --warmup 2 --no-cpu-baseline --emulate-world 8 --emulate-rank $r --headline $mode 2>/dev/null \
 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode\t8\t$r\t%d\t%.3f\t%.3f\t%.4g' % (sum(v['batches'] for v in d['scan_launches'].values()), d['ms_per_step'], d['rank0_ms']['kernels_total'], d['value']))" >> gpurun_out/r04/emulate_full_collection_n8.tsv
done; done
cat gpurun_out/r04/emulate_full_collection_n8.tsv
