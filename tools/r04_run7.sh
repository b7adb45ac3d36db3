timeout 900 python -m pytest tests/test_gpu_runs.py tests/test_gpu_cli.py -m gpu -x -q -k "rccl or eight or two_ranks" 2>&1 | tail -5
BENCH_FORCE_DIST=1 python3 bench.py --steps 3 --warmup 1 --rows-divisor 400 --queries 3000 --no-cpu-baseline --only-headline 2>/dev/null | cut -c1-100
