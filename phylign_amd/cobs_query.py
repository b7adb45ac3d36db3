"""`cobs query`-compatible command line on top of libphylign_match.so.

Accepts the argv the reference passes to the external binary
(scripts/run_cobs_streaming.sh:24-29, Snakefile:419-424, Snakefile:476-481):

    cobs query [--load-complete] -t THRESH -T THREADS -i INDEX [--index-sizes BYTES] -f QUERY.fa

and prints the same result text on stdout.  INDEX may be a regular file or a
pipe such as /dev/fd/63.  `--load-complete` and `-T` are accepted and ignored
(the index is always fully resident in HBM; results never depended on the
thread count).  Extensions: `--nb-best-hits N` fuses `postprocess_cobs.py -n N`
(Snakefile:425 / :467) into the same process; `--device D` picks the GPU;
`--server SOCK[,SOCK...]` (or PHYLIGN_MATCH_SERVER) sends the job to a resident-index
server (phylign_amd/server.py; with one socket per GPU a batch always goes to the same one) -- then INDEX must be a path the server can open
(the .xz itself is fine, the server decodes it once and keeps it in HBM).
Any failure exits non-zero with the reason on stderr (Snakefile:142 runs rules
under `set -euo pipefail`; scripts/benchmark.py:56-61 re-raises).
"""
import argparse
import os
import sys


def build_parser():
    ap = argparse.ArgumentParser(prog="cobs", description="MI355X-native `cobs query` drop-in")
    sub = ap.add_subparsers(dest="cmd", required=True)
    q = sub.add_parser("query", help="query a COBS classic index")
    q.add_argument("-i", "--index", dest="index", required=True, action="append",
                   help="path to a .cobs_classic index (file or pipe); may be given once")
    q.add_argument("-f", "--file", dest="query_file", required=True, help="query FASTA (single-line, upper-case ACGT)")
    q.add_argument("-t", "--threshold", type=float, default=0.8, help="fraction of k-mers that must match [0.8]")
    q.add_argument("-T", "--threads", type=int, default=0, help="accepted for compatibility; ignored")
    q.add_argument("-l", "--limit", type=int, default=0, help="number of results to return per query, 0 = all")
    q.add_argument("--load-complete", action="store_true", help="accepted for compatibility (always true here)")
    q.add_argument("--index-sizes", type=int, nargs="*", default=[], help="size of the index stream in bytes")
    q.add_argument("--nb-best-hits", type=int, default=None,
                   help="fuse scripts/postprocess_cobs.py -n N into the output")
    q.add_argument("--device", type=int, default=0, help="GPU ordinal")
    q.add_argument("--server", default=os.environ.get("PHYLIGN_MATCH_SERVER"),
                   help="unix socket of a resident-index server")
    return ap


def pick_server(servers, index_path):
    """`--server` may list one socket per GPU, comma separated: a batch always goes to the same one
    (CRC-32 of its file name), so each server of an 8-GPU node keeps its own eighth of the 661k
    indexes resident -- the whole database (1.06 TB decompressed) fits 8 x 288 GB, not one GPU."""
    import zlib
    socks = [s for s in servers.split(",") if s]
    if len(socks) <= 1:
        return socks[0] if socks else servers
    name = os.path.basename(index_path)
    for ext in (".xz", ".cobs_classic", ".cobs_compact"):
        if name.endswith(ext):
            name = name[: -len(ext)]
    return socks[zlib.crc32(name.encode()) % len(socks)]


def run_query(args, out=None):
    from . import _lib as pm
    out = out or sys.stdout.buffer
    if len(args.index) != 1:
        raise SystemExit("error: exactly one -i index is supported (Phylign passes one per batch)")
    if args.limit and (args.server or args.nb_best_hits is not None):
        raise SystemExit("error: -l/--limit cannot be combined with --server or --nb-best-hits")
    size_hint = args.index_sizes[0] if args.index_sizes else 0
    if args.server:
        from .server import request
        # the server runs on this host (unix socket): it is told where the query file is instead of being sent its
        # bytes; a pipe or anything else that cannot be re-opened is sent
        req = {"op": "query", "index": os.path.abspath(args.index[0]), "index_size": size_hint,
               "threshold": args.threshold, "nb_best_hits": args.nb_best_hits}
        fasta = b""
        if os.path.isfile(args.query_file):
            req["fasta_path"] = os.path.abspath(args.query_file)
        else:
            with open(args.query_file, "rb") as f:
                fasta = f.read()
            req["fasta_len"] = len(fasta)
        head, body = request(pick_server(args.server, args.index[0]), req, fasta)
        if not head.get("ok"):
            raise RuntimeError(head.get("error", "server error"))
        out.write(body)
        out.flush()
        return
    pm.init(args.device)
    ix = pm.Index.load_file(args.index[0], size_hint=size_hint)
    with open(args.query_file, "rb") as f:
        fasta = f.read()
    nb = -1 if args.nb_best_hits is None else max(args.nb_best_hits, 0)
    if args.limit > 0:
        # -l N: the GPU keeps the N best + ties per query, the formatter prints exactly N
        q = pm.Queries(fasta, term_size=ix.info.term_size)
        res = pm.search([ix], q, args.threshold, nb_best_hits=args.limit)
        out.write(pm.format_hits_limit(ix, q, res.hits(), limit=args.limit))
        res.free()
    else:
        out.write(pm.query_text(ix, fasta, args.threshold, nb_best_hits=nb))
    out.flush()
    ix.free()


def main(argv=None):
    args = build_parser().parse_args(argv)
    try:
        run_query(args)
    except SystemExit:
        raise
    except Exception as e:          # PMError, OSError, ImportError (library not built)
        print(f"cobs query (phylign_amd): {e}", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
