#!/usr/bin/env bash
# PMC passes (each in its own run, --kernel-trace only) for the bench workload and
# the calibration kernels.  Run on the GPU box from the repo root:
#   bash tools/run_pmc.sh <tag>     -> gpurun_out/pmc_<tag>/...   (headline = fetch-all scan)
#   PMC_BENCH_FLAGS="--headline threshold_bound" bash tools/run_pmc.sh <tag>_bound
set -u
tag=${1:-r01}
out=gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --only-headline --no-pipeline ${PMC_BENCH_FLAGS:-}"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o bench -- $B > $out/bench_fetch.json 2> $out/bench_fetch.log
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/rdreq -o bench -- $B > $out/bench_rdreq.json 2> $out/bench_rdreq.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum --output-format csv -d $out/hit -o bench -- $B > $out/bench_hit.json 2> $out/bench_hit.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o bench -- $B > $out/bench_write.json 2> $out/bench_write.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/calib_fetch -o calib -- python3 tools/pmc_calib.py > $out/calib_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/calib_rdreq -o calib -- python3 tools/pmc_calib.py > $out/calib_rdreq.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/calib_write -o calib -- python3 tools/pmc_calib.py > $out/calib_write.log 2>&1
find $out -name "*.csv" | head -30
du -sh $out
