#!/usr/bin/env bash
# Read-request size split of the narrow-row gather (run on the GPU box from the repo root):
#   bash tools/run_narrow_pmc.sh r03   -> gpurun_out/narrow_<tag>/{timing.txt,pmc_*.txt}
set -u
tag=${1:-r03}
out=gpurun_out/narrow_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/narrow_calib.py > $out/timing.txt 2> $out/timing.err
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/size -o nc -- python3 tools/narrow_calib.py --pmc > $out/size.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_READ_SECTORS_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/req -o nc -- python3 tools/narrow_calib.py --pmc > $out/req.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUBBLE_sum TCC_TAG_STALL_sum --output-format csv -d $out/stall -o nc -- python3 tools/narrow_calib.py --pmc > $out/stall.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_BUSY_sum --output-format csv -d $out/dram -o nc -- python3 tools/narrow_calib.py --pmc > $out/dram.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o nc -- python3 tools/narrow_calib.py --pmc > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_REQ_sum --output-format csv -d $out/tcp -o nc -- python3 tools/narrow_calib.py --pmc > $out/tcp.log 2>&1
for d in size req stall dram fetch tcp; do python3 tools/pmc_summary.py $out/$d > $out/pmc_$d.txt 2>&1; done
find $out -name "*.csv" -size +1M -delete
ls -la $out
