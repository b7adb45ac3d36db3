#!/usr/bin/env bash
# Round profile set on the GPU box (run from the repo root through gpurun):
#   bash tools/run_profiles.sh r02
# -> gpurun_out/prof_<tag>/{fetch_all,bound}/ kernel traces with --stats, the bench line of each run,
#    and the PMC passes of tools/run_pmc.sh for both scan modes.
set -u
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/fetch_all -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-pmc --whole-record --only-headline > $out/bench_fetch_all.json 2> $out/bench_fetch_all.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bound -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-pmc --whole-record --only-headline --headline threshold_bound > $out/bench_bound.json 2> $out/bench_bound.log
# the gene-length leg (the length mix of data/ARGannot_r3.fa): the 10- and 13-plane instantiations beside the headline's
rocprofv3 --kernel-trace --stats --output-format csv -d $out/argannot -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-live-pmc --whole-record --no-clustered --no-l31 --no-full-shard --no-unique-rows > $out/bench_argannot.json 2> $out/bench_argannot.log
bash tools/run_pmc.sh ${tag}
PMC_BENCH_FLAGS="--headline threshold_bound" bash tools/run_pmc.sh ${tag}_bound
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}/fetch gpurun_out/pmc_${tag}/rdreq gpurun_out/pmc_${tag}/hit gpurun_out/pmc_${tag}/write > $out/pmc_fetch_all.txt
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_bound/fetch gpurun_out/pmc_${tag}_bound/rdreq gpurun_out/pmc_${tag}_bound/hit gpurun_out/pmc_${tag}_bound/write > $out/pmc_bound.txt
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}/calib_fetch gpurun_out/pmc_${tag}/calib_rdreq gpurun_out/pmc_${tag}/calib_write > $out/pmc_calibration.txt
python3 tools/make_pmc_traffic.py $out/pmc_fetch_all.txt $out/pmc_bound.txt > $out/pmc_traffic.json
find $out -name "*stats*.csv" | head
# drop the bulky per-dispatch traces, keep the summaries
find gpurun_out/pmc_${tag} gpurun_out/pmc_${tag}_bound -name "*.csv" -size +2M -delete
