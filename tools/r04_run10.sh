set -x
mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests_final.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04/gputests_final.log
tail -4 gpurun_out/r04/gputests_final.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_n1_driver_flags.json 2> gpurun_out/r04/bench_n1_driver_flags.log ) 2>&1 | tail -3
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 python3 bench.py --gpus 8 --steps 2 --rows-divisor 400 --no-cpu-baseline > gpurun_out/r04/bench_plain_gpus8_gloo_shared.json 2> gpurun_out/r04/bench_plain_gpus8_gloo_shared.log; echo "plain --gpus 8 rc=$?"
bash tools/run_profiles.sh r04 > gpurun_out/r04/run_profiles.log 2>&1
ls gpurun_out/prof_r04
head -c 600 gpurun_out/prof_r04/pmc_traffic.json
