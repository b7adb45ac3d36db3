for rep in 1 2; do
for fd in "" 1; do
for mode in fetch_all_rows threshold_bound; do
for r in 1 4; do
BENCH_FORCE_DIST=$fd python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --emulate-world 8 --emulate-rank $r --headline $mode 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('rccl_in_loop=${fd:-0} $mode rank $r  %.3f ms/step  kernels %.3f  gather %.3f depth %s' % (d['ms_per_step'], d['rank0_ms']['kernels_total'], d['rank0_ms']['hit_gather'], d['config'].get('pipeline_depth')))"
done; done; done; done
