#!/usr/bin/env bash
# SQ wait / issue counters of the scan kernels in both scan modes (one rocprofv3 --pmc pass per mode, GRBM_GUI_ACTIVE in a
# pass of its own).  GPU box, repo root:  bash tools/run_sq.sh r03  -> gpurun_out/sq_<tag>/pmc_sq_scan.txt
set -u
tag=${1:-r03}
out=gpurun_out/sq_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --only-headline --no-pipeline"
{
echo "# rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
echo "#   -- $B --headline <mode>   (one pass per mode; GRBM_GUI_ACTIVE in a pass of its own)"
echo "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are in quad-cycles summed over waves (MI355X_MICROARCH.md)"
for mode in fetch_all_rows threshold_bound; do
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
      --output-format csv -d $out/sq_$mode -o bench -- $B --headline $mode > /dev/null 2> $out/sq_$mode.log
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/grbm_$mode -o bench -- $B --headline $mode > /dev/null 2> $out/grbm_$mode.log
  echo "## $mode"
  python3 tools/pmc_summary.py $out/sq_$mode $out/grbm_$mode | grep -v "^#" | grep "k_scan\|dispatches"
done
} > $out/pmc_sq_scan.txt
find $out -name "*.csv" -size +1M -delete
cat $out/pmc_sq_scan.txt
