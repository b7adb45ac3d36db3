set -x
mkdir -p gpurun_out/r04
df -h /tmp . | tee gpurun_out/r04/df.txt
nproc; cat /sys/fs/cgroup/cpu.max
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests_2.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04/gputests_2.log
tail -5 gpurun_out/r04/gputests_2.log
free_gb=$(df --output=avail -BG /tmp | tail -1 | tr -dc 0-9)
div=16; [ "$free_gb" -gt 90 ] && div=8
timeout 1500 python3 tools/e2e_cold_warm.py --rows-divisor $div --queries 100000 --work /tmp/cw --out gpurun_out/r04/cold_warm_div${div}_100k.json 2> gpurun_out/r04/cold_warm_err.log | tail -c 3000
tail -5 gpurun_out/r04/cold_warm_err.log
