mkdir -p gpurun_out/r04
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r04/gputests_final.log 2>&1; echo "tests rc=$?" >> gpurun_out/r04/gputests_final.log
tail -4 gpurun_out/r04/gputests_final.log
python3 -c "import __graft_entry__ as g; g.smoke()"
