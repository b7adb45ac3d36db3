mkdir -p gpurun_out/r04
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_n1_driver_flags.json 2> gpurun_out/r04/bench_n1_driver_flags.log ) 2>&1 | tail -3
python3 -c "
import json; d=json.loads(open('gpurun_out/r04/bench_n1_driver_flags.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline'])"
