mkdir -p gpurun_out/r04
timeout 2400 bash tools/emulate_scaling.sh "1 2 4 8" > gpurun_out/r04/emulate_scaling_n1_2_4_8.tsv 2> gpurun_out/r04/emulate_err2.log
cat gpurun_out/r04/emulate_scaling_n1_2_4_8.tsv
