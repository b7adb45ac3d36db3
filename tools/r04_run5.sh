set -x
mkdir -p gpurun_out/r04
timeout 600 python -m pytest tests/test_gpu_runs.py -m gpu -x -q -k "saved_index" 2>&1 | tail -5
timeout 900 python3 tools/e2e_full_collection.py --rows-divisor 64 --queries 1000000 --work /tmp/fc --out gpurun_out/r04/full_collection_8ranks_div64_1M.json > gpurun_out/r04/full_collection.log 2> gpurun_out/r04/full_collection.err; echo "rc=$?"
tail -5 gpurun_out/r04/full_collection.err
python3 - <<'P'
import json
d=json.load(open('gpurun_out/r04/full_collection_8ranks_div64_1M.json'))
print(json.dumps(d['runs'],indent=0)[:3000])
P
