for sl in 0 1; do
PM_SINGLE_LAUNCH=$sl python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --only-headline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=1 single_launch=$sl  %.3f ms/step' % d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['scan_launches'].items()})"
for r in 0 1 4; do
PM_SINGLE_LAUNCH=$sl python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --emulate-world 8 --emulate-rank $r 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=8 r=$r single_launch=$sl  %.3f ms/step' % d['ms_per_step'], {k:round(v['avg_ms'],3) for k,v in d['scan_launches'].items()})"
done; done
