#!/usr/bin/env bash
# Round 6's recorded runs, on the final source (GPU box, repo root):  bash tools/r06_numbers.sh
#   gpurun_out/final_r06/: the driver's command (line + side file), the plain 8-rank invocation at full size over gloo on the
#   one GPU (line + side file: the size cap with 8 ranks), the emulated N-way split.
set -u
out=gpurun_out/final_r06
mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 --legs-out $out/bench_n1_driver_flags_legs.json > $out/bench_n1_driver_flags.json 2> $out/bench_n1_driver_flags.err; echo "bench n1 rc=$? line $(wc -c < $out/bench_n1_driver_flags.json) bytes"
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 BENCH_FULL_MIN_WORLD=99 python3 bench.py --gpus 8 --steps 5 --warmup 1 --legs-out $out/bench_plain_gpus8_gloo_shared_gpu_fullsize_legs.json \
    > $out/bench_plain_gpus8_gloo_shared_gpu_fullsize.json 2> $out/bench_plain_gpus8_gloo_shared_gpu_fullsize.err; echo "bench n8 (gloo, one GPU) rc=$? line $(wc -c < $out/bench_plain_gpus8_gloo_shared_gpu_fullsize.json) bytes"
BENCH_DIST_BACKEND=gloo BENCH_SHARE_GPU=1 BENCH_FULL_MIN_WORLD=8 python3 bench.py --gpus 8 --steps 2 --warmup 1 --rows-divisor 400 --queries 3000 --cpu-target-s 0.6 --cpu-sample-gb 0.2 \
    --legs-out $out/bench_plain_gpus8_gloo_shared_gpu_div400_full_collection_legs.json > $out/bench_plain_gpus8_gloo_shared_gpu_div400_full_collection.json 2> $out/bench_plain_gpus8_div400.err
echo "bench n8 + full_collection (rows/400) rc=$? line $(wc -c < $out/bench_plain_gpus8_gloo_shared_gpu_div400_full_collection.json) bytes"
bash tools/emulate_scaling.sh "1 2 4 8" --no-live-pmc > $out/emulate_scaling_n1_2_4_8.tsv 2> $out/emulate_scaling.err
cat $out/emulate_scaling_n1_2_4_8.tsv
