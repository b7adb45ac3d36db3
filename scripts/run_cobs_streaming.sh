#!/usr/bin/env bash
# Drop-in for the reference's scripts/run_cobs_streaming.sh (same five positional
# arguments, same stdout): the xz-compressed classic index is streamed through a
# pipe into the MI355X matching stage instead of the `cobs` binary.
#   usage: run_cobs_streaming.sh kmer_thres threads cobs_index.xz uncompressed_size query.fa
set -e
set -o pipefail
set -u

prog=$(basename "$0")
here=$(cd "$(dirname "$0")" && pwd)
if [[ $# -ne 5 ]]; then
	>&2 echo "usage: $prog kmer_thres threads cobs_index.xz uncompressed_size query.fa"
	exit 1
fi
thres="$1"; nthreads="$2"; index_xz="$3"; index_bytes="$4"; fasta="$5"

export PYTHONPATH="${here}/..${PYTHONPATH:+:$PYTHONPATH}"
if [[ -n "${PHYLIGN_MATCH_SERVER:-}" ]]; then
	# resident-index server: it decodes the .xz once and keeps the matrix in HBM
	exec python3 -m phylign_amd.cobs_query query --load-complete -t "${thres}" -T "${nthreads}" \
		-i "${index_xz}" --index-sizes "${index_bytes}" -f "${fasta}" --server "${PHYLIGN_MATCH_SERVER}" \
		${PHYLIGN_NB_BEST_HITS:+--nb-best-hits "${PHYLIGN_NB_BEST_HITS}"}
fi
exec python3 -m phylign_amd.cobs_query query --load-complete \
	-t "${thres}" \
	-T "${nthreads}" \
	-i <(xzcat --no-sparse --ignore-check "${index_xz}") \
	--index-sizes "${index_bytes}" \
	-f "${fasta}" \
	${PHYLIGN_NB_BEST_HITS:+--nb-best-hits "${PHYLIGN_NB_BEST_HITS}"} \
	${PHYLIGN_GPU:+--device "${PHYLIGN_GPU}"}
