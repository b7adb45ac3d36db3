set -x
mkdir -p gpurun_out/r04
df -h /tmp | tail -1
timeout 1300 python3 tools/e2e_full_collection.py --rows-divisor 32 --queries 1000000 --work /tmp/fc --out gpurun_out/r04/full_collection_8ranks_div32_1M.json > gpurun_out/r04/full_collection.log 2> gpurun_out/r04/full_collection.err; echo "rc=$?"
tail -5 gpurun_out/r04/full_collection.err
