for rep in 1 2; do
for d in 1 2 3; do
for mode in fetch_all_rows threshold_bound; do
for r in 1 5; do
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --emulate-world 8 --emulate-rank $r --headline $mode --pipeline-depth $d 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('depth $d $mode rank $r  %.3f ms/step  kernels %.3f' % (d['ms_per_step'], d['rank0_ms']['kernels_total']))"
done; done; done; done
for d in 1 2; do
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-headline --pipeline-depth $d 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=1 depth $d  %.3f ms/step  kernels %.3f' % (d['ms_per_step'], d['rank0_ms']['kernels_total']))"
done
