"""04_filter consumer of the 03_match files: for every query keep the globally
best n matches (+ ties with the n-th) across all batch files and emit
">qname ref1,ref2,...\\nseq".  Behavioural mirror of the reference's
scripts/filter_queries.py:27-206, pinned by
tests/golden/filter/*.  Reads .gz or plain text match files.
"""
import argparse
import gzip
import os
import sys
from collections import OrderedDict


def _open_text(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "rt")


def read_queries(path):
    """FASTA/FASTQ reader: name = first word of the header (reference readfq, :69-103)."""
    out = OrderedDict()
    with _open_text(path) as f:
        lines = [ln.rstrip("\n") for ln in f]
    i, n = 0, len(lines)
    while i < n:
        while i < n and lines[i][:1] not in (">", "@"):
            i += 1
        if i >= n:
            break
        fastq = lines[i][0] == "@"
        name = lines[i][1:].partition(" ")[0]
        i += 1
        seq = []
        while i < n and lines[i][:1] not in ("@", "+", ">"):
            seq.append(lines[i])
            i += 1
        seq = "".join(seq)
        if i < n and lines[i][:1] == "+":       # FASTQ quality block: skip len(seq) characters
            i += 1
            got = 0
            while i < n and got < len(seq):
                got += len(lines[i])
                i += 1
        elif fastq:
            pass
        out[name] = seq
    return out


def iter_match_file(path):
    """yields (qname, [(ref, kmers), ...]) per '*' record of one 03_match file
    (grammar: reference :46-66: header '*qname[ comment]\\tN', hit '<rid>_<ref>\\t<kmers>')."""
    qname, hits, seen = None, [], False
    with _open_text(path) as f:
        for raw in f:
            line = raw.strip()
            if not line:
                continue
            if line[0] == "*":
                if seen:
                    yield qname, hits
                    hits = []                      # (only here: hit lines ahead of the first header stay in the
                head = line[1:].split("\t")        #  buffer and join the first query, as in the reference :52-56)
                qname = head[0].split(" ")[0]
                int(head[1])                       # must parse, like the reference
                seen = True
            else:
                name, kmers = line.split()         # exactly two whitespace-separated fields
                _rid, ref = name.split("_")        # exactly one underscore
                hits.append((ref, int(kmers)))
    if not seen:
        raise ValueError(f"{path}: no '*' query header in match file")
    yield qname, hits


class TopMatches:
    """best `keep` (+ties) of one query across batches; order (-kmers, batch, ref)"""

    def __init__(self, keep):
        self.keep = keep
        self.floor = 0
        self.items = []          # (batch, ref, kmers)

    def add(self, batch, hits):
        for ref, kmers in hits:
            if kmers >= self.floor:
                self.items.append((batch, ref, kmers))
        self.items.sort(key=lambda t: (-t[2], t[0], t[1]))
        if len(self.items) > self.keep:
            cut = self.keep
            self.floor = self.items[cut - 1][2]
            while cut < len(self.items) and self.items[cut][2] == self.floor:
                cut += 1
            del self.items[cut:]


def filter_files(query_path, match_paths, keep, out, log=None):
    seqs = read_queries(query_path)
    best = OrderedDict((name, TopMatches(keep)) for name in seqs)
    for path in match_paths:
        batch = os.path.basename(str(path)).split("____")[0]
        if log:
            print(f"Translating matches {path}", file=log)
        for qname, hits in iter_match_file(path):
            if qname not in best:
                raise KeyError(f"{path}: query '{qname}' is not in {query_path}")
            best[qname].add(batch, hits)
    for name, top in best.items():
        out.write(f">{name} {','.join(t[1] for t in top.items)}\n{seqs[name]}\n")


def filter_files_native(query_path, match_paths, keep, out_bin, log=None):
    """the same job through the native merge (pm_merge_add_text / pm_merge_emit: host code of
    libphylign_match.so, no GPU involved) -- minutes instead of hours at a million reads x 305 files.
    Returns False when the query file is not the prepared single-line ACGT FASTA the native reader takes
    (then the caller uses filter_files)."""
    from . import _lib as pm
    with (gzip.open(query_path, "rb") if str(query_path).endswith(".gz") else open(query_path, "rb")) as f:
        fasta = f.read()
    try:
        q = pm.Queries(fasta, term_size=1)
    except pm.PMError:
        return False
    if q.count()[0] != sum(1 for ln in fasta.split(b"\n") if ln[:1] in (b">", b"@")):
        return False                               # records without sequence, FASTQ ...: the general reader handles them
    m = pm.Merge(q, keep)

    def one(path):
        batch = os.path.basename(str(path)).split("____")[0]
        with (gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")) as f:
            m.add_text(batch, f.read())           # inflating and parsing run outside the GIL; the merge serialises its own part

    if log:
        for path in match_paths:
            print(f"Translating matches {path}", file=log)
    # the result does not depend on the order of the files (the best n + ties of a union), so they are read in parallel
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(8, len(os.sched_getaffinity(0)), len(match_paths)))) as ex:
        list(ex.map(one, match_paths))
    out_bin.write(m.emit())
    return True


def main(argv=None):
    ap = argparse.ArgumentParser(description="keep the n best COBS matches per query across batches")
    ap.add_argument("match_fn", nargs="+")
    ap.add_argument("-q", dest="query_fn", required=True, help="query file")
    ap.add_argument("-n", dest="keep", type=int, default=100, help="no. of best hits to keep [100]")
    ap.add_argument("--python", action="store_true", help="use the pure-Python reader even when the native library is built")
    a = ap.parse_args(argv)
    if not a.python:
        try:
            from . import _lib as pm
            pm.load()
        except (ImportError, OSError):
            pm = None
        if pm is not None:
            try:
                sys.stdout.flush()
                if filter_files_native(a.query_fn, a.match_fn, a.keep, sys.stdout.buffer, log=sys.stderr):
                    sys.stdout.buffer.flush()
                    return
            except pm.PMError as e:                 # a malformed match file, an unknown query: the rule fails
                print(f"filter_queries (phylign_amd): {e}", file=sys.stderr)
                sys.exit(1)
    filter_files(a.query_fn, a.match_fn, a.keep, sys.stdout, log=sys.stderr)


if __name__ == "__main__":
    main()
