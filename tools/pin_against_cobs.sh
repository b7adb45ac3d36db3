#!/usr/bin/env bash
# Turns "parity unpinned" into "pinned": captures what a REAL `cobs` binary (bioconda cobs=0.2.1, the version
# /root/reference/envs/cobs.yaml:5 pins) prints for the reference's exact command line, as fixtures under
# tests/golden/cobs/, and runs the test that compares the oracle and the HIP product with them.
#
#   conda create -n cobs -c bioconda cobs=0.2.1 && conda activate cobs      (any machine with network)
#   bash tools/pin_against_cobs.sh                                          (repo root; no GPU needed for the capture)
#   python -m pytest tests/test_cobs_pin.py -q                              (CPU: the checker; with -m gpu on an MI355X: the product)
#
# What is captured (all small, committed as data):
#   tests/golden/cobs/genomes/*.fa          24 genome-like strains of one species (tools/make_pin_inputs.py, seeded)
#   tests/golden/cobs/index.cobs_classic    `cobs classic-construct` of them  -> pins the header byte layout
#   tests/golden/cobs/queries.fa            400 reads: 151 bp (121 k-mers: ceil(0.7 x 121) = 85, floor = 84), 150 bp, 31-40 bp,
#                                           error rates 0-12 % so that scores land on both sides of the threshold and
#                                           many documents tie
#   tests/golden/cobs/cobs_stdout.txt       stdout of the reference's argv (scripts/run_cobs_streaming.sh:24-29, Snakefile:419-424)
#   tests/golden/cobs/cobs_stdout_stream.txt  the same through a pipe with --index-sizes (mem-stream mode)
#   tests/golden/cobs/edge_*.txt            stdout / exit status for a read shorter than k and a read with an N
#   tests/golden/cobs/cobs_version.txt
# The test then says which setting of the two switchable rules (cobs_threshold_rule x cobs_tie_order) reproduces the
# text byte for byte; if it is not the default, change the defaults in pm_runtime.cpp and in the checker under tests (one line each).
set -euo pipefail
cd "$(dirname "$0")/.."
G=tests/golden/cobs
command -v cobs >/dev/null || { echo "no 'cobs' on PATH: conda install -c bioconda cobs=0.2.1" >&2; exit 2; }
ver=$(cobs version 2>&1 | head -3 | tr '\n' ' ')
echo "$ver" | grep -q "0\.2\.1" || { echo "cobs version is '$ver', the reference pins 0.2.1 (envs/cobs.yaml:5); set PIN_ANY_VERSION=1 to go on" >&2; [ -n "${PIN_ANY_VERSION:-}" ] || exit 2; }
mkdir -p "$G"
echo "$ver" > "$G/cobs_version.txt"
python3 tools/make_pin_inputs.py "$G"
rm -f "$G/index.cobs_classic"
# 661k parameters: k = 31, one hash function, false positive rate 0.3, canonical k-mers
cobs classic-construct --term-size 31 --num-hashes 1 --false-positive-rate 0.3 --file-type fasta --clobber "$G/genomes" "$G/index.cobs_classic" \
  || cobs classic-construct -k 31 --num-hashes 1 --false-positive-rate 0.3 "$G/genomes" "$G/index.cobs_classic"
# the reference's argv, on-disk form (Snakefile:419-424) ...
cobs query --load-complete -t 0.7 -T 1 -i "$G/index.cobs_classic" -f "$G/queries.fa" > "$G/cobs_stdout.txt"
# ... and the streaming form (scripts/run_cobs_streaming.sh:24-29)
size=$(stat -c %s "$G/index.cobs_classic")
cobs query --load-complete -t 0.7 -T 2 -i <(cat "$G/index.cobs_classic") --index-sizes "$size" -f "$G/queries.fa" > "$G/cobs_stdout_stream.txt"
cobs query --load-complete -t 0.0 -T 1 -i "$G/index.cobs_classic" -f "$G/queries_few.fa" > "$G/cobs_stdout_t0.txt"
for e in short with_n; do
  set +e
  cobs query --load-complete -t 0.7 -T 1 -i "$G/index.cobs_classic" -f "$G/edge_$e.fa" > "$G/edge_$e.stdout.txt" 2> "$G/edge_$e.stderr.txt"
  echo $? > "$G/edge_$e.status.txt"
  set -e
done
ls -la "$G" | head -30
python3 -m pytest tests/test_cobs_pin.py -q -m "not gpu"
echo "captured.  Commit tests/golden/cobs/ and run 'python -m pytest tests/test_cobs_pin.py -m gpu' on an MI355X."
