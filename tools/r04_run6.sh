set -x
mkdir -p gpurun_out/r04
python3 -c "import torch, torch.distributed" 
timeout 900 python -m pytest tests/test_gpu_runs.py tests/test_gpu_cli.py -m gpu -x -q -k "rccl or eight or two_ranks" 2>&1 | tail -5
timeout 900 python3 tools/e2e_full_collection.py --rows-divisor 32 --queries 1000000 --work /tmp/fc --out gpurun_out/r04/full_collection_8ranks_div32_1M.json > gpurun_out/r04/full_collection.log 2> gpurun_out/r04/full_collection.err; echo "rc=$?"
tail -5 gpurun_out/r04/full_collection.err
