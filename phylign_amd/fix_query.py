"""Query preprocessing ahead of the matching stage: behavioural mirror of rules
`fix_query` and `concatenate_queries` (Snakefile:314-352), i.e. of
`seqtk seq -A -U -C in | awk 'odd lines as is; even lines: gsub(/[^ACGT]/, "A")'`:
FASTA or FASTQ in (multi-line allowed), one `>name` line (comment dropped) and
one upper-case sequence line with every non-ACGT byte replaced by `A` out.
seqtk is not available in the build container, so this mirror is pinned only
by the awk half (tests/test_golden_cpu.py::test_fix_query_*); SURVEY.md 8f rank 4.
"""
import gzip
import sys

_KEEP = bytes(65 if (c not in b"ACGT") else c for c in range(256))      # after upper-casing


def _open(path):
    return gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")


def iter_records(stream):
    """kseq-style FASTA/FASTQ reader: yields (name, sequence bytes)"""
    name, seq, in_qual, qual_left = None, [], False, 0
    for raw in stream:
        line = raw.rstrip(b"\r\n")
        if in_qual:
            qual_left -= len(line)
            if qual_left <= 0:
                in_qual = False
            continue
        if line[:1] in (b">", b"@"):
            if name is not None:
                yield name, b"".join(seq)
            name, seq = line[1:].split(None, 1)[0] if line[1:].split() else b"", []
        elif line[:1] == b"+" and name is not None:
            qual_left = sum(len(s) for s in seq)
            in_qual = qual_left > 0
        elif name is not None:
            seq.append(line)
    if name is not None:
        yield name, b"".join(seq)


def fix_stream(stream, out, base_to_replace=b"A"):
    table = _KEEP if base_to_replace == b"A" else bytes((base_to_replace[0] if (c not in b"ACGT") else c) for c in range(256))
    for name, seq in iter_records(stream):
        out.write(b">" + name + b"\n" + seq.upper().translate(table) + b"\n")


def fix_files(paths, out):
    """fix_query on every input + concatenate_queries in the given order"""
    for p in paths:
        with _open(p) as f:
            fix_stream(f, out)


def main(argv=None):
    paths = (argv if argv is not None else sys.argv[1:]) or ["-"]
    for p in paths:
        if p == "-":
            fix_stream(sys.stdin.buffer, sys.stdout.buffer)
        else:
            fix_files([p], sys.stdout.buffer)
    sys.stdout.buffer.flush()


if __name__ == "__main__":
    main()
