"""ctypes front-end of oracle/libcobs_oracle.so (see cobs_oracle.h).

TEST INFRASTRUCTURE ONLY -- "parity unpinned" against bioconda cobs=0.2.1 (the
binary and its sources are absent from /root/reference); XXH64 is pinned by
tests/golden/xxh64_kat.tsv.  Nothing under phylign_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcobs_oracle.so")


class Header(C.Structure):
    _fields_ = [("version", C.c_uint32), ("term_size", C.c_uint32),
                ("canonicalize", C.c_uint8), ("signature_size", C.c_uint64),
                ("num_hashes", C.c_uint64), ("n_docs", C.c_uint32),
                ("row_bytes", C.c_uint64), ("names_off", C.c_size_t),
                ("data_off", C.c_size_t), ("layout", C.c_int)]


class Compact(C.Structure):
    _fields_ = [("term_size", C.c_uint32), ("canonicalize", C.c_uint8), ("n_parts", C.c_uint32),
                ("n_docs", C.c_uint32), ("page_size", C.c_uint64), ("params_off", C.c_size_t),
                ("names_off", C.c_size_t), ("data_off", C.c_size_t)]


class Hit(C.Structure):
    _fields_ = [("doc", C.c_uint32), ("score", C.c_uint32)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_xxh64.restype = C.c_uint64
        L.orc_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.orc_canonicalize.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_threshold.restype = C.c_uint32
        L.orc_threshold.argtypes = [C.c_double, C.c_uint64]
        L.orc_header_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Header)]
        L.orc_index_alloc.restype = C.c_void_p
        L.orc_index_alloc.argtypes = [C.c_uint32, C.c_uint8, C.c_uint64, C.c_uint64, C.c_uint32,
                                      C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_create_hashes.argtypes = [C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.c_uint64, C.c_void_p]
        L.orc_scores.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Header), C.c_char_p, C.c_size_t, C.c_void_p]
        L.orc_select.restype = C.c_size_t
        L.orc_select.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.c_size_t, C.c_void_p]
        L.orc_query_file.restype = C.c_void_p
        L.orc_query_file.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_double, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t]
        L.orc_compact_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Compact)]
        L.orc_compact_alloc.restype = C.c_void_p
        L.orc_compact_alloc.argtypes = [C.c_uint32, C.c_uint8, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                        C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_scores_compact.argtypes = [C.c_void_p, C.POINTER(Compact), C.c_char_p, C.c_size_t, C.c_void_p]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_splitmix64.argtypes = [C.c_uint64]
        L.orc_synth_row.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p]
        L.orc_synth_fill.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_baseline_run.restype = C.c_uint64
        L.orc_baseline_run.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Header), C.c_char_p, C.c_size_t,
                                       C.c_size_t, C.c_double, C.c_int]
        L.orc_baseline_run_slabs.restype = C.c_uint64
        L.orc_baseline_run_slabs.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(Header), C.c_char_p, C.c_size_t,
                                             C.c_size_t, C.c_double, C.c_int, C.c_size_t]
        _lib = L
    return _lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def xxh64(data: bytes, seed: int = 0) -> int:
    return lib().orc_xxh64(data, len(data), seed)


def canonicalize(kmer: bytes):
    out = C.create_string_buffer(len(kmer))
    ok = lib().orc_canonicalize(kmer, len(kmer), out)
    return out.raw if ok else None


def threshold(t: float, num_terms: int) -> int:
    return lib().orc_threshold(t, num_terms)


def set_rules(threshold_rule=0, tie_order=0):
    """the two rules of `cobs query` no reference file pins: threshold_rule 0 = ceil(t * k-mers) (default), 1 = floor,
    2 = round half up; tie_order 0 = equal scores by ascending document index (default), 1 = descending.  The product's
    pm_set_option("cobs_threshold_rule" / "cobs_tie_order") take the same values."""
    lib().orc_set_rules(int(threshold_rule), int(tie_order))


def header_parse(buf) -> Header:
    import numpy as np
    a = np.frombuffer(buf, dtype=np.uint8)
    h = Header()
    if lib().orc_header_parse(a.ctypes.data, a.size, C.byref(h)) != 0:
        raise ValueError("not a COBS classic index")
    return h


def make_index(term_size, canonicalize, signature_size, num_hashes, names, matrix=None):
    """Build a complete .cobs_classic image (bytes-like numpy array).
    matrix: optional uint8 array [signature_size, ceil(n_docs/8)]."""
    import numpy as np
    n = len(names)
    arr = (C.c_char_p * max(n, 1))(*[s.encode() for s in names])
    tot, off = C.c_size_t(), C.c_size_t()
    p = lib().orc_index_alloc(term_size, canonicalize, signature_size, num_hashes, n, arr, C.byref(tot), C.byref(off))
    if not p:
        raise MemoryError
    buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(tot.value,)).copy()
    _libc.free(p)
    if matrix is not None:
        m = np.ascontiguousarray(matrix, dtype=np.uint8)
        assert m.size == tot.value - off.value, (m.shape, tot.value - off.value)
        buf[off.value:] = m.reshape(-1)
    return buf


def make_compact(term_size, canonicalize, page_size, sig_sizes, num_hashes, names, matrices=None):
    """Build a complete .cobs_compact image.  matrices: optional list of uint8 arrays
    [sig_sizes[p], page_size] (sub-index p holds documents p*page_size*8 ...)."""
    import numpy as np
    n, parts = len(names), len(sig_sizes)
    arr = (C.c_char_p * max(n, 1))(*[s.encode() for s in names])
    sig = np.asarray(sig_sizes, dtype=np.uint64)
    nh = np.asarray(num_hashes, dtype=np.uint64)
    tot, off = C.c_size_t(), C.c_size_t()
    p = lib().orc_compact_alloc(term_size, canonicalize, page_size, parts, sig.ctypes.data, nh.ctypes.data, n, arr,
                                C.byref(tot), C.byref(off))
    if not p:
        raise MemoryError
    buf = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(tot.value,)).copy()
    _libc.free(p)
    if matrices is not None:
        o = off.value
        for m, s in zip(matrices, sig_sizes):
            m = np.ascontiguousarray(m, dtype=np.uint8)
            assert m.shape == (s, page_size)
            buf[o:o + m.size] = m.reshape(-1)
            o += m.size
    return buf


def compact_parse(buf) -> Compact:
    import numpy as np
    a = np.frombuffer(buf, dtype=np.uint8)
    c = Compact()
    if lib().orc_compact_parse(a.ctypes.data, a.size, C.byref(c)) != 0:
        raise ValueError("not a COBS compact index")
    return c


def create_hashes(seq: bytes, k: int, canon: int, num_hashes: int):
    import numpy as np
    nt = len(seq) - k + 1
    out = np.zeros(max(nt, 0) * num_hashes, dtype=np.uint64)
    rc = lib().orc_create_hashes(seq, len(seq), k, canon, num_hashes, out.ctypes.data)
    if rc:
        raise ValueError(f"orc_create_hashes rc={rc}")
    return out


def scores(index_buf, seq: bytes):
    import numpy as np
    a = np.frombuffer(index_buf, dtype=np.uint8)
    h = header_parse(a)
    sc = np.zeros(h.n_docs, dtype=np.uint32)
    rc = lib().orc_scores(a.ctypes.data + h.data_off, h.row_bytes, C.byref(h), seq, len(seq), sc.ctypes.data)
    if rc:
        raise ValueError(f"orc_scores rc={rc}")
    return sc


def scores_rows(rows_fn, h: Header, seq: bytes):
    """Scores against a *virtual* matrix: rows_fn(row_index) -> uint8 array of
    h.row_bytes bytes.  Pure numpy; used for sampled parity at full sizes."""
    import numpy as np
    hs = create_hashes(seq, h.term_size, h.canonicalize, h.num_hashes).reshape(-1, h.num_hashes)
    sc = np.zeros(h.n_docs, dtype=np.uint32)
    for term in hs:
        acc = None
        for hv in term:
            r = np.asarray(rows_fn(int(hv) % h.signature_size), dtype=np.uint8)
            acc = r if acc is None else (acc & r)
        sc += np.unpackbits(acc, bitorder="little")[: h.n_docs]
    return sc


def select(sc, num_terms: int, thr: float, num_results: int = 0):
    import numpy as np
    sc = np.ascontiguousarray(sc, dtype=np.uint32)
    hits = (Hit * max(len(sc), 1))()
    n = lib().orc_select(sc.ctypes.data, len(sc), num_terms, thr, num_results, hits)
    return [(hits[i].doc, hits[i].score) for i in range(n)]


def query_file(index_buf, fasta: bytes, thr: float, num_results: int = 0) -> bytes:
    import numpy as np
    a = np.frombuffer(index_buf, dtype=np.uint8)
    n = C.c_size_t()
    err = C.create_string_buffer(256)
    p = lib().orc_query_file(a.ctypes.data, a.size, fasta, len(fasta), thr, num_results, C.byref(n), err, 256)
    if not p:
        raise RuntimeError(err.value.decode())
    out = C.string_at(p, n.value)
    _libc.free(p)
    return out


def splitmix64(x: int) -> int:
    return lib().orc_splitmix64(x)


def synth_row(seed: int, batch: int, row: int, n_docs: int):
    import numpy as np
    out = np.zeros((n_docs + 7) // 8, dtype=np.uint8)
    lib().orc_synth_row(seed, batch, row, n_docs, out.ctypes.data)
    return out


def synth_matrix(seed: int, batch: int, n_rows: int, n_docs: int):
    import numpy as np
    rb = (n_docs + 7) // 8
    m = np.zeros((n_rows, rb), dtype=np.uint8)
    for r in range(n_rows):
        lib().orc_synth_row(seed, batch, r, n_docs, m[r].ctypes.data)
    return m


def synth_fill(seed: int, batch: int, n_rows: int, n_docs: int, threads: int = 1):
    """whole synthetic matrix [n_rows, ceil(n_docs/8)] built with `threads` threads"""
    import numpy as np
    rb = (n_docs + 7) // 8
    m = np.zeros((n_rows, rb), dtype=np.uint8)
    lib().orc_synth_fill(seed, batch, n_rows, n_docs, rb, m.ctypes.data, threads)
    return m


def baseline_run(matrix, stride, h: Header, seqs: bytes, qlen: int, n_queries: int, thr: float, threads: int) -> int:
    return lib().orc_baseline_run(matrix.ctypes.data, stride, C.byref(h), seqs, qlen, n_queries, thr, threads)


def baseline_run_slabs(matrix, stride, h: Header, seqs: bytes, qlen: int, n_queries: int, thr: float, threads: int,
                       slab_bytes: int = 64) -> int:
    """baseline_run with the work split into column slabs the way `cobs query -T` splits it"""
    return lib().orc_baseline_run_slabs(matrix.ctypes.data, stride, C.byref(h), seqs, qlen, n_queries, thr, threads, slab_bytes)
