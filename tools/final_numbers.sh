#!/usr/bin/env bash
# The round's reported numbers in one go (GPU box, repo root):  bash tools/final_numbers.sh r03
#   gpurun_out/final_<tag>/bench_n1_driver_flags.json (the driver's flags) + e2e_*.json (tools/e2e_numbers.sh)
set -u
tag=${1:-r03}
out=gpurun_out/final_$tag
mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 --legs-out $out/bench_n1_driver_flags_legs.json > $out/bench_n1_driver_flags.json 2> $out/bench.log; echo "bench rc=$?"
bash tools/e2e_numbers.sh $out
