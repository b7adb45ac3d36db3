"""Post-filter of the 03_match stage: keep the n best hits per query plus every
later hit tied with the n-th, and cut the random sorting prefix off document
names.  Behavioural mirror of the reference's scripts/postprocess_cobs.py:10-39
(same stdin->stdout contract, same quirks, same failure cases); pinned by
tests/golden/postprocess/*.  The fused native form is pm_format_hits()
(include/phylign_match.h) -- this module is for text that already exists.
"""
import argparse
import sys


class PostprocessError(Exception):
    pass


def _score(line):
    # last TAB-separated field as int (reference :10-13); raises ValueError on "_"
    return int(line.rsplit("\t", 1)[-1])


def _strip_prefix(line):
    # everything up to and including the first '_' of the LINE becomes '_'
    # (reference :16-18); a line without '_' collapses to a bare "_" with no newline
    cut = line.find("_")
    return "_" + (line[cut + 1:] if cut >= 0 else "")


def filter_stream(lines, keep, out):
    """lines: iterable of text lines (with their newlines); keep: -n value."""
    rank = None          # hits seen for the current query; None before the first header
    nth_score = 0
    for line in lines:
        if line[:1] == "*":
            out.write(line)
            rank, nth_score = 0, 0
            continue
        if rank is None:
            raise PostprocessError("hit line before the first '*' header")
        short = _strip_prefix(line)
        rank += 1
        if rank < keep:
            out.write(short)
        elif rank == keep:
            out.write(short)
            nth_score = _score(short)
        elif _score(short) == nth_score:
            out.write(short)


def filter_text(text, keep):
    import io
    out = io.StringIO()
    filter_stream(io.StringIO(text), keep, out)
    return out.getvalue()


def main(argv=None):
    ap = argparse.ArgumentParser(description="Postprocess cobs output: keep top n hits (+ties) and remove random identifiers")
    ap.add_argument("-n", metavar="int", dest="keep", required=True, type=int, help="no. of best hits to keep")
    args = ap.parse_args(argv)
    try:
        filter_stream(sys.stdin, args.keep, sys.stdout)
    finally:
        sys.stdout.flush()


if __name__ == "__main__":
    main()
