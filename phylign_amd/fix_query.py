"""Query preprocessing ahead of the matching stage: behavioural mirror of rules
`fix_query` and `concatenate_queries` (Snakefile:314-352), i.e. of
`seqtk seq -A -U -C in | awk 'odd lines as is; even lines: gsub(/[^ACGT]/, "A")'`:
FASTA or FASTQ in (multi-line allowed), one `>name` line (comment dropped) and
one upper-case sequence line with every non-ACGT byte replaced by `A` out.
seqtk is not available in the build container: its reader (kseq.h's kseq_read) is
restated step by step in iter_records, the awk half is pinned by running awk
(tests/test_golden_cpu.py::test_fix_query_*); SURVEY.md 8f rank 4.
"""
import gzip
import sys

_KEEP = bytes(65 if (c not in b"ACGT") else c for c in range(256))      # after upper-casing


def _open(path):
    return gzip.open(path, "rb") if str(path).endswith(".gz") else open(path, "rb")


_SPACE = b" \t\n\v\f\r"                                   # isspace(): what ends a record's name in kseq


def iter_records(stream):
    """FASTA/FASTQ records as kseq.h's kseq_read() delivers them to seqtk -- the steps of that function restated on a
    byte string: yields (name, sequence bytes); a FASTQ record whose quality is truncated or overlong ends the input
    (seqtk's read loop stops at kseq_read() < 0) and is not yielded"""
    data = stream if isinstance(stream, (bytes, bytearray)) else stream.read()
    n, p, last = len(data), 0, 0
    while True:
        if last == 0:                                      # jump to the next header byte, wherever in a line it stands
            hits = [i for i in (data.find(b">", p), data.find(b"@", p)) if i >= 0]
            if not hits:
                return
            p = min(hits) + 1
        if p >= n:
            return
        e = p
        while e < n and data[e] not in _SPACE:
            e += 1
        name, seq, c = data[p:e], bytearray(), -1
        if e < n:
            p = e + 1
            if data[e] != 10:                              # the comment: to the end of the line
                nl = data.find(b"\n", p)
                p = n if nl < 0 else nl + 1
            while p < n:                                   # sequence lines
                c = data[p]
                if c in b">+@":
                    p += 1
                    break
                if c == 10:
                    p, c = p + 1, -1
                    continue
                nl = data.find(b"\n", p)
                end = n if nl < 0 else nl
                seq += data[p:end]
                p = n if nl < 0 else nl + 1
                if len(seq) > 1 and seq[-1] == 13:
                    del seq[-1]
                c = -1
        else:
            p = n
        if c in (62, 64):
            last = c
        if c != 43:                                        # FASTA record (or the input ended)
            yield bytes(name), bytes(seq)
            if c == -1:
                return
            continue
        nl = data.find(b"\n", p)                           # FASTQ: the rest of the '+' line, then the quality
        if nl < 0:
            return
        p, qual = nl + 1, 0
        while p < n:
            nl = data.find(b"\n", p)
            end = n if nl < 0 else nl
            ln = end - p
            if qual + ln > 1 and ln and data[end - 1] == 13:
                ln -= 1
            qual += ln
            p = n if nl < 0 else nl + 1
            if not qual < len(seq):
                break
        last = 0
        if qual != len(seq):
            return
        yield bytes(name), bytes(seq)


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
