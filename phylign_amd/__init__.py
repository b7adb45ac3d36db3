"""phylign_amd -- MI355X-native COBS k-mer matching stage (intermediate/03_match)
for the Phylign pipeline.  The compute path is libphylign_match.so (hand-written
HIP for gfx950) behind the C ABI in include/phylign_match.h; this package is the
Python host layer that mirrors the reference's `run_cobs_streaming.sh` /
`cobs query` / `postprocess_cobs.py` interface.  There is no CPU fallback."""
__version__ = "0.1.0"
