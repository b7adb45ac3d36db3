#!/usr/bin/env bash
# Wide differential sweep of the HIP path against the oracle: N extra seeded cases of each fuzz test
# (random shapes / k / hash counts / thresholds / formats; several batches per search in every scan form).
#   bash tools/fuzz_sweep.sh [N=1500] [first seed=100]      (GPU box; ~0.1 s per case)
set -uo pipefail
cd "$(dirname "$0")/.."
PM_FUZZ_EXTRA=${1:-1500} PM_FUZZ_OFFSET=${2:-100} python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
