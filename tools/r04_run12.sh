mkdir -p gpurun_out/r04
timeout 1500 bash tools/fuzz_sweep.sh 1500 9000 > gpurun_out/r04/fuzz_sweep_seeds_9000.txt 2>&1
tail -5 gpurun_out/r04/fuzz_sweep_seeds_9000.txt
timeout 900 bash tools/e2e_numbers.sh gpurun_out/r04/e2e > gpurun_out/r04/e2e_numbers.txt 2>&1
cat gpurun_out/r04/e2e_numbers.txt | cut -c1-700
