"""661k-shaped synthetic workloads and the static batch->GPU map (8e).

The shape table phylign_amd/data/shapes_661k.tsv is derived from the
reference's data/661k_batches.txt.xz and data/decompressed_indexes_sizes.txt by
tools/make_shape_table.py; nothing here reads /root/reference at run time.
"""
import os
from collections import namedtuple

import numpy as np

Shape = namedtuple("Shape", "batch n_docs index_bytes row_bytes signature_size in_small in_config3 batch_id")
_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "shapes_661k.tsv")


def load_shapes(path=_TABLE):
    out = []
    with open(path) as f:
        for line in f:
            if line.startswith("#") or not line.strip():
                continue
            b, d, nbytes, rb, s, sm, c3 = line.rstrip("\n").split("\t")
            out.append(Shape(b, int(d), int(nbytes), int(rb), int(s), bool(int(sm)), bool(int(c3)), len(out)))
    return out


def select(workload, shapes=None):
    """'config3': the 64 batches of BASELINE.json configs[2] (every 5th of the
    sorted full list + the 3 batches_small batches, ~213 GB);
    'small': the 3 batches of data/batches_small.txt; 'full': all 305;
    'config2': bacillus_anthracis__01 only (BASELINE configs[1])."""
    shapes = shapes or load_shapes()
    if workload == "config3":
        return [s for s in shapes if s.in_config3]
    if workload == "small":
        return [s for s in shapes if s.in_small]
    if workload == "config2":
        return [s for s in shapes if s.batch == "bacillus_anthracis__01"]
    if workload == "full":
        return list(shapes)
    raise ValueError(workload)


def scale_shapes(shapes, rows_divisor):
    """same document counts (same algorithmic bytes per k-mer), fewer rows"""
    return [s._replace(signature_size=max(1024, s.signature_size // rows_divisor),
                       index_bytes=max(1024, s.signature_size // rows_divisor) * s.row_bytes) for s in shapes]


# measured scan time per looked-up row by 128-byte lines per row, relative to 400 for a 4-line row
# (tools/per_batch_cost.py on one MI355X, 12 M lookups per batch, fetch-all scan with non-temporal gathers;
# profiles/r03/per_batch_cost.tsv): 0.281 / 0.511 / 0.726 / 0.966 ms for 1 / 2 / 3 / 4 lines
# plus ~25 units per batch whatever its width: in the fused launches of an 8-way split a rank with more (narrow) batches
# ran 0.06-0.08 ms per extra batch behind the others (profiles/r03/emulate_scaling.tsv)
_LINE_COST = {1: 141, 2: 237, 3: 326, 4: 425}


def scan_cost(shape):
    """relative scan cost of one query k-mer against this batch: HBM moves whole
    128-byte lines, so a row costs its width rounded up to lines,
    weighted by the measured per-line cost of that row width"""
    lines = (shape.row_bytes + 127) // 128
    return _LINE_COST.get(lines, 100 * lines)


def assign_batches(shapes, n_ranks, capacity_bytes=None):
    """Static batch -> rank map: greedy longest-processing-time on scan_cost (128-B
    lines per looked-up row), ties broken by index bytes; optional per-rank
    capacity on the sum of index bytes.  Returns a list of lists of positions
    into `shapes`, each sorted ascending."""
    order = sorted(range(len(shapes)), key=lambda i: (-scan_cost(shapes[i]), -shapes[i].index_bytes, i))
    load = [0] * n_ranks
    used = [0] * n_ranks
    out = [[] for _ in range(n_ranks)]
    for i in order:
        cands = sorted(range(n_ranks), key=lambda r: (load[r], used[r], r))
        for r in cands:
            if capacity_bytes is None or used[r] + shapes[i].index_bytes <= capacity_bytes:
                break
        else:
            raise MemoryError(f"batch {shapes[i].batch} does not fit on any rank")
        out[r].append(i)
        load[r] += scan_cost(shapes[i])
        used[r] += shapes[i].index_bytes
    # greedy LPT leaves the heaviest rank a whole small batch above the others (4 % at 8 ranks x 64 batches): refine by
    # moving one batch, or exchanging two, between the heaviest rank and any other while that lowers the pair's maximum
    cost = [scan_cost(s) for s in shapes]
    nbytes = [s.index_bytes for s in shapes]
    fits = (lambda r, delta: True) if capacity_bytes is None else (lambda r, delta: used[r] + delta <= capacity_bytes)
    for _ in range(4 * len(shapes)):
        a = max(range(n_ranks), key=lambda r: (load[r], -r))
        best = None                                            # (new pair maximum, b, item of a, item of b or None)
        for b in range(n_ranks):
            if b == a:
                continue
            for i in out[a]:
                if fits(b, nbytes[i]):                          # move i: a -> b
                    m = max(load[a] - cost[i], load[b] + cost[i])
                    if m < load[a] and (best is None or (m, b, i, -1) < (best[0], best[1], best[2], -1 if best[3] is None else best[3])):
                        best = (m, b, i, None)
                for j in out[b]:                                # swap i <-> j
                    if cost[j] >= cost[i] or not fits(b, nbytes[i] - nbytes[j]) or not fits(a, nbytes[j] - nbytes[i]):
                        continue
                    m = max(load[a] - cost[i] + cost[j], load[b] + cost[i] - cost[j])
                    if m < load[a] and (best is None or (m, b, i, j) < (best[0], best[1], best[2], -1 if best[3] is None else best[3])):
                        best = (m, b, i, j)
        if best is None:
            break
        _, b, i, j = best
        out[a].remove(i); out[b].append(i)
        load[a] -= cost[i]; load[b] += cost[i]; used[a] -= nbytes[i]; used[b] += nbytes[i]
        if j is not None:
            out[b].remove(j); out[a].append(j)
            load[b] -= cost[j]; load[a] += cost[j]; used[b] -= nbytes[j]; used[a] += nbytes[j]
    return [sorted(x) for x in out]


PART_DEN = 1024          # query shares are expressed in 1/1024 of the query set (pm_qpart_t.den)
# the part of scan_cost a batch costs whatever share of the queries it is searched with.  Measured on one MI355X
# (profiles/r04/NOTES.md): none that the run-to-run noise lets one see -- ranks with 9-10 batches of an 8-way split of
# config 3 finish with those that hold 7-8.
_BATCH_FIXED = int(os.environ.get("PHYLIGN_BATCH_FIXED", "0"))


def assign_parts(shapes, n_ranks, capacity_bytes=None, max_replica_bytes=4 << 30, tolerance=0.004):
    """assign_batches, then the residual imbalance (whole batches: the heaviest of 8 ranks is left ~4 % above the mean on the
    64 batches of config 3) is levelled by making a few SMALL batches resident on two ranks that share their queries:
    the heaviest rank hands the upper end of one batch's query range to the lightest one (pm_search_async_parts), sized so
    that one of them reaches the mean under cost(share) = _BATCH_FIXED + share * (scan_cost - _BATCH_FIXED).  Only batches of at most
    max_replica_bytes are replicated (the narrow one-line batches; a replica costs its HBM a second time).
    Returns per rank a list of (position in shapes, lo, hi) sorted by position: the rank searches the batch with the
    queries of share [lo, hi) / PART_DEN; (pos, 0, PART_DEN) is a whole batch."""
    whole = assign_batches(shapes, n_ranks, capacity_bytes)
    held = [{pos: [0, PART_DEN] for pos in part} for part in whole]
    if n_ranks < 2:
        return [[(pos, 0, PART_DEN) for pos in part] for part in whole]
    var = [max(1, scan_cost(s) - _BATCH_FIXED) for s in shapes]

    def load_of(r):
        return sum(_BATCH_FIXED + var[p] * (hi - lo) / PART_DEN for p, (lo, hi) in held[r].items())

    def used_of(r):
        return sum(shapes[p].index_bytes for p in held[r])
    for _ in range(4 * n_ranks):
        load = [load_of(r) for r in range(n_ranks)]
        a = max(range(n_ranks), key=lambda r: (load[r], -r))
        b = min(range(n_ranks), key=lambda r: (load[r], r))
        # the mean every rank should reach once b paid the fixed cost of one more resident batch; the transfer brings the
        # heaviest rank down to it or the lightest one up to it, whichever is less (n - 1 transfers level n ranks)
        target = (sum(load) + _BATCH_FIXED) / n_ranks
        amount = min(load[a] - target, target - load[b] - _BATCH_FIXED)
        if amount <= tolerance * target:
            # nobody is far enough below the mean to take a share AND its fixed cost up to the mean: level the pair instead
            amount = (load[a] - load[b] - _BATCH_FIXED) / 2.0
            if amount <= tolerance * target:
                break                                                                # not worth a second resident copy
        best = None
        for p, (lo, hi) in held[a].items():
            if p in held[b] or hi - lo != PART_DEN or shapes[p].index_bytes > max_replica_bytes:
                continue                                                             # a batch is shared by two ranks at most
            if capacity_bytes is not None and used_of(b) + shapes[p].index_bytes > capacity_bytes:
                continue
            x = int(round(amount / var[p] * PART_DEN))
            if not 0 < x < PART_DEN:
                continue
            key = (shapes[p].index_bytes, p)                                         # the cheapest replica that can carry it
            if best is None or key < best[0]:
                best = (key, p, x)
        if best is None:
            break
        _, p, x = best
        lo, hi = held[a][p]
        held[a][p] = [lo, hi - x]
        held[b][p] = [hi - x, hi]
    return [sorted((p, lo, hi) for p, (lo, hi) in h.items()) for h in held]


def parts_cost(shapes, parts):
    """modelled scan cost of every rank of an assign_parts() result"""
    return [sum(_BATCH_FIXED + max(1, scan_cost(shapes[p]) - _BATCH_FIXED) * (hi - lo) / PART_DEN for p, lo, hi in part)
            for part in parts]


def assign_named(batches, sizes, n_ranks, capacity_bytes=None):
    """assign_batches for a real batch list (match_stage): `batches` are names, `sizes` maps a name to
    its decompressed index bytes (data/decompressed_indexes_sizes.txt).  A batch of the 661k collection
    takes its document count -- hence its scan cost -- from the shape table; an unknown batch is priced
    by the row width its size implies at the collection's typical 12.8 M rows."""
    table = {s.batch: s for s in load_shapes()}
    shapes = []
    for i, b in enumerate(batches):
        nbytes = int(sizes.get(b, 0))
        if b in table:
            s = table[b]
            shapes.append(s._replace(index_bytes=nbytes or s.index_bytes, batch_id=i))
        else:
            rb = max(1, nbytes // 12_800_000)
            shapes.append(Shape(b, rb * 8, max(nbytes, 1), rb, max(1, nbytes // rb), False, False, i))
    return assign_batches(shapes, n_ranks, capacity_bytes)


def make_queries(n, length=150, seed=31):
    """uniform ACGT queries; returns (fasta bytes, uint8 array [n, length])"""
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=(n, length), dtype=np.uint8)
    seqs = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    hdr = [f">q{i:07d}\n".encode() for i in range(n)]
    parts = []
    for i in range(n):
        parts.append(hdr[i])
        parts.append(seqs[i].tobytes())
        parts.append(b"\n")
    return b"".join(parts), seqs


_ARGANNOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "argannot_lengths.txt")


def argannot_lengths(path=_ARGANNOT):
    """sequence lengths of the 1 856 records of the reference's data/ARGannot_r3.fa in file order (SURVEY.md 8d's third
    query shape: genes of 237 ... 3 153 bp, 1 594 532 31-mers); package data, lengths only
    (written by tools/gen_golden_argannot.py; tests/test_golden_cpu.py re-derives it from the reference when that is present)"""
    with open(path) as f:
        return [int(x) for x in f if x.strip() and not x.startswith("#")]


def make_queries_lengths(lengths, seed=31, prefix="g"):
    """uniform ACGT queries of the given lengths, in that order; returns (fasta bytes, list of sequences as bytes)"""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    seqs = [lut[rng.integers(0, 4, size=int(n), dtype=np.uint8)].tobytes() for n in lengths]
    parts = []
    for i, s_ in enumerate(seqs):
        parts.append(f">{prefix}{i:07d}\n".encode())
        parts.append(s_)
        parts.append(b"\n")
    return b"".join(parts), seqs


def plant_plan_ragged(hashes, terms, shapes, every=20, docs_per_query=8, seed=7, threshold=0.7):
    """plant_plan for queries of different lengths: `terms` = k-mers per query, `hashes` = their seed-0 hashes one query
    after the other.  The planted fractions straddle the threshold: 1.0, 0.9, 0.8, threshold, threshold - 0.05, 0.6."""
    rng = np.random.default_rng(seed)
    fr = [1.0, 0.9, 0.8, threshold, threshold - 0.05, 0.6]
    off = np.concatenate([[0], np.cumsum(np.asarray(terms, dtype=np.int64))])
    plan, sure = {}, 0
    for n, q in enumerate(range(0, len(terms), every)):
        pos = n % len(shapes)
        sh = shapes[pos]
        hq = hashes[off[q]:off[q + 1]]
        rows_q = (hq % np.uint64(sh.signature_size)).astype(np.uint64)
        docs = rng.choice(sh.n_docs, size=min(docs_per_query, sh.n_docs), replace=False)
        for j, d in enumerate(docs):
            m = int(np.ceil(fr[j % len(fr)] * terms[q]))
            r, dd = plan.setdefault(pos, ([], []))
            r.append(rows_q[:m])
            dd.append(np.full(m, d, dtype=np.uint32))
            sure += int(m >= np.ceil(threshold * terms[q]))
    return {p_: (np.concatenate(r), np.concatenate(d)) for p_, (r, d) in plan.items()}, sure


def plant_plan(hashes, n_queries, terms_per_query, shapes, every=20, docs_per_query=8, seed=7):
    """Planted true positives: every `every`-th query is planted into one batch
    (round-robin over `shapes`) in `docs_per_query` documents at match fractions
    cycling through 1.0, 0.9, 0.8, 0.7, 0.65, 0.6.  hashes: uint64 [n_queries*terms]
    (seed-0 hashes, num_hashes = 1).  Returns {shape position: (rows, docs)} and the
    number of (query, doc) pairs whose planted fraction alone reaches 0.7."""
    rng = np.random.default_rng(seed)
    fr = [1.0, 0.9, 0.8, 0.7, 0.65, 0.6]
    plan = {}
    sure = 0
    h = hashes.reshape(n_queries, terms_per_query)
    for n, q in enumerate(range(0, n_queries, every)):
        pos = n % len(shapes)
        sh = shapes[pos]
        rows_q = (h[q] % np.uint64(sh.signature_size)).astype(np.uint64)
        docs = rng.choice(sh.n_docs, size=min(docs_per_query, sh.n_docs), replace=False)
        for j, d in enumerate(docs):
            f = fr[j % len(fr)]
            m = int(np.ceil(f * terms_per_query))
            r, dd = plan.setdefault(pos, ([], []))
            r.append(rows_q[:m])
            dd.append(np.full(m, d, dtype=np.uint32))
            sure += int(m >= np.ceil(0.7 * terms_per_query))
    return {p: (np.concatenate(r), np.concatenate(d)) for p, (r, d) in plan.items()}, sure
