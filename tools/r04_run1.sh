set -x
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests_1.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04/gputests_1.log
tail -5 gpurun_out/r04/gputests_1.log
bash tools/emulate_scaling.sh "1 8" > gpurun_out/r04/emulate_parts_fixed25.tsv 2> gpurun_out/r04/emulate_err.log
PHYLIGN_BATCH_FIXED=0 bash tools/emulate_scaling.sh "8" > gpurun_out/r04/emulate_parts_fixed0.tsv 2>> gpurun_out/r04/emulate_err.log
bash tools/emulate_scaling.sh "8" --no-replicas > gpurun_out/r04/emulate_whole.tsv 2>> gpurun_out/r04/emulate_err.log
cat gpurun_out/r04/emulate_parts_fixed25.tsv gpurun_out/r04/emulate_parts_fixed0.tsv gpurun_out/r04/emulate_whole.tsv
