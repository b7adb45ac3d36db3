"""ctypes binding of libphylign_match.so (include/phylign_match.h).

This is the reference-side stub a Phylign maintainer would add to call the
MI355X stage instead of spawning `cobs query`
(scripts/run_cobs_streaming.sh:24-29).  Loading fails loudly when the HIP
extension has not been built; compute calls fail loudly without a gfx950 GPU.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PHYLIGN_MATCH_LIB: another build of the same library (e.g. the host code under AddressSanitizer for the CPU tests)
LIB_PATH = os.environ.get("PHYLIGN_MATCH_LIB") or os.path.join(_HERE, "libphylign_match.so")

PM_LAYOUT_AUTO, PM_LAYOUT_COMPACT, PM_LAYOUT_ALIGNED = 0, 1, 2
PM_DOC_COUNT = 0xFFFFFFFF      # doc value of a "count record" (see include/phylign_match.h)
ERR_NAMES = {-1: "PM_EINVAL", -2: "PM_ENODEV", -3: "PM_ENOMEM", -4: "PM_EIO",
             -5: "PM_EFORMAT", -6: "PM_EQUERY", -7: "PM_EHIP", -8: "PM_ERANGE"}


class PMError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class IndexInfo(C.Structure):
    _fields_ = [("term_size", C.c_uint32), ("canonicalize", C.c_uint32),
                ("signature_size", C.c_uint64), ("num_hashes", C.c_uint32),
                ("n_docs", C.c_uint32), ("row_bytes", C.c_uint64), ("stride", C.c_uint64),
                ("device_bytes", C.c_uint64), ("header_layout", C.c_uint32), ("has_matrix", C.c_uint32),
                ("n_parts", C.c_uint32), ("reserved", C.c_uint32), ("page_size", C.c_uint64)]


class Stats(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("n_terms", C.c_uint64), ("n_hits", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("ms_total", C.c_double), ("ms_hash", C.c_double),
                ("ms_scan", C.c_double), ("n_scan_launches", C.c_uint32), ("reserved", C.c_uint32),
                ("n_records", C.c_uint64), ("n_runs", C.c_uint64), ("fetched_bytes", C.c_uint64)]


class Launch(C.Structure):
    _fields_ = [("lanes_per_row", C.c_uint32), ("planes", C.c_uint32), ("num_hashes", C.c_uint32),
                ("n_batches", C.c_uint32), ("n_queries", C.c_uint64), ("algorithmic_bytes", C.c_uint64),
                ("ms", C.c_double), ("fetched_bytes", C.c_uint64), ("wide_query", C.c_uint32), ("reserved", C.c_uint32)]


HIT_DTYPE = np.dtype([("query", "<u4"), ("doc", "<u4"), ("score", "<u4"), ("slot", "<u4")])

# every symbol include/phylign_match.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("pm_init", C.c_int, [C.c_int]),
    ("pm_shutdown", None, []),
    ("pm_last_error", C.c_char_p, []),
    ("pm_device_info", C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    ("pm_free", None, [_P]),
    ("pm_set_option", C.c_int, [C.c_char_p, C.c_int64]),
    ("pm_threshold_terms", C.c_uint32, [C.c_double, C.c_uint64]),
    ("pm_index_load_file", C.c_int, [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(_P)]),
    ("pm_index_load_fd", C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(_P)]),
    ("pm_index_load_mem", C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    ("pm_index_load_fd_tee", C.c_int, [C.c_int, C.c_uint64, C.c_int, C.c_char_p, C.POINTER(C.c_int), C.POINTER(_P)]),
    ("pm_index_load_header_mem", C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    ("pm_index_create", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_char_p, C.c_size_t, C.c_uint32, C.c_int, C.c_int, C.POINTER(_P)]),
    ("pm_index_matrix_device", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("pm_index_from_names", C.c_int, [C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    ("pm_index_drop_matrix", C.c_int, [_P]),
    ("pm_index_read_rows", C.c_int, [_P, C.c_uint64, C.c_uint64, _P]),
    ("pm_index_info", C.c_int, [_P, C.POINTER(IndexInfo)]),
    ("pm_index_device", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("pm_index_doc_name", _P, [_P, C.c_uint32, C.POINTER(C.c_size_t)]),
    ("pm_index_read_row", C.c_int, [_P, C.c_uint64, _P]),
    ("pm_index_free", None, [_P]),
    ("pm_queries_parse", C.c_int, [C.c_char_p, C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    ("pm_queries_parse_raw", C.c_int, [_P, C.c_size_t, C.c_uint32, C.c_int, C.POINTER(_P)]),
    ("pm_fasta_record_cuts", C.c_int, [_P, C.c_size_t, C.c_uint64, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("pm_queries_fasta", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("pm_queries_count", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("pm_queries_terms", C.c_int, [_P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("pm_queries_free", None, [_P]),
    ("pm_queries_release_device", C.c_int, [_P]),
    ("pm_queries_device_bytes", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("pm_hash_terms", C.c_int, [_P, C.c_int, C.c_uint32, _P]),
    ("pm_search", C.c_int, [C.POINTER(_P), C.c_size_t, _P, C.c_double, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    ("pm_search_async", C.c_int, [C.POINTER(_P), C.c_size_t, _P, C.c_double, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    ("pm_search_async_parts", C.c_int, [C.POINTER(_P), C.c_size_t, _P, C.c_double, C.c_uint32, C.c_uint32, _P, C.POINTER(_P)]),
    ("pm_result_wait", C.c_int, [_P]),
    ("pm_result_stats", C.c_int, [_P, C.POINTER(Stats)]),
    ("pm_result_launches", C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("pm_hits_sort", None, [_P, C.c_uint64]),
    ("pm_result_hits_device", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("pm_result_ordered_device", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("pm_result_copy_hits_device", C.c_int, [_P, _P, C.c_uint64, C.c_int]),
    ("pm_result_hits_into", C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("pm_result_hits_host", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("pm_result_free", None, [_P]),
    ("pm_format_hits", C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_int64, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("pm_format_hits_limit", C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("pm_format_hits_gz", C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_int64, C.c_char_p, C.c_int,
                                    C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("pm_result_slot_hits", C.c_int, [_P, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint64), C.POINTER(_P)]),
    ("pm_slice_free", None, [_P]),
    ("pm_gzip_fast", C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("pm_format_hits_gz_piece", C.c_int, [_P, _P, _P, C.c_uint64, C.c_uint32, C.c_int64, C.c_char_p, C.c_int, C.c_int,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("pm_merge_create", C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    ("pm_merge_add", C.c_int, [_P, C.c_char_p, _P, _P, C.c_uint64, C.c_uint32, C.c_int64]),
    ("pm_merge_add_text", C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_size_t]),
    ("pm_merge_export", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    ("pm_merge_emit", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("pm_merge_extend", C.c_int, [_P, _P]),
    ("pm_merge_add_piece", C.c_int, [_P, C.c_int64, C.c_char_p, _P, _P, C.c_uint64, C.c_uint32, C.c_int64]),
    ("pm_merge_batches", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("pm_merge_emit_file", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_uint64)]),
    ("pm_merge_emit_file_piece", C.c_int, [_P, C.c_char_p, C.c_int, C.POINTER(C.c_uint64)]),
    ("pm_merge_free", None, [_P]),
    ("pm_query_text", C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_double, C.c_int64, C.POINTER(_P), C.POINTER(C.c_size_t)]),
]

_lib = None


def load():
    """dlopen the HIP library; raises (never falls back) when it is missing.

    A process that also drives the GPU through torch (bench.py, the multi-rank match_stage)
    must `import torch` BEFORE the first call of this function: the torch wheel bundles its own
    libamdhip64 under the same SONAME as /opt/rocm's, the dynamic loader keeps whichever came
    first for both users, and torch finds no GPU on the other one."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the gfx950 HIP extension first "
                "(python -m phylign_amd.build or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _chk(rc):
    if rc != 0:
        raise PMError(rc, load().pm_last_error().decode(errors="replace"))


_inited_device = None


def init(device=0):
    global _inited_device
    _chk(load().pm_init(device))
    _inited_device = device


def shutdown():
    """releases the library's streams and pooled buffers (init() may be called again)"""
    global _inited_device
    load().pm_shutdown()
    _inited_device = None


def bound_device():
    """GPU ordinal given to init() (None before)"""
    return _inited_device


def set_option(name, value):
    _chk(load().pm_set_option(name.encode(), int(value)))


def device_info():
    name = C.create_string_buffer(256)
    tot, fr, cus = C.c_uint64(), C.c_uint64(), C.c_int()
    _chk(load().pm_device_info(name, 256, C.byref(tot), C.byref(fr), C.byref(cus)))
    return {"name": name.value.decode(), "hbm_total": tot.value, "hbm_free": fr.value, "cus": cus.value}


def threshold_terms(threshold, num_terms):
    return load().pm_threshold_terms(threshold, num_terms)


class Index:
    """One phylogenetic batch index resident in HBM (or header-only)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load_file(cls, path, size_hint=0, layout=PM_LAYOUT_AUTO):
        h = _P()
        _chk(load().pm_index_load_file(os.fsencode(path), size_hint, layout, C.byref(h)))
        return cls(h)

    @classmethod
    def load_fd(cls, fd, size_hint=0, layout=PM_LAYOUT_AUTO, tee_path=None):
        """tee_path: also keep the stream as that file (decode-once cache); the index's `cached` says whether it exists"""
        h = _P()
        if tee_path is None:
            _chk(load().pm_index_load_fd(fd, size_hint, layout, C.byref(h)))
            return cls(h)
        cached = C.c_int(0)
        _chk(load().pm_index_load_fd_tee(fd, size_hint, layout, os.fsencode(tee_path), C.byref(cached), C.byref(h)))
        ix = cls(h)
        ix.cached = bool(cached.value)
        return ix

    @classmethod
    def load_mem(cls, buf, layout=PM_LAYOUT_AUTO):
        a = np.frombuffer(buf, dtype=np.uint8)
        h = _P()
        _chk(load().pm_index_load_mem(a.ctypes.data, a.size, layout, C.byref(h)))
        return cls(h)

    @classmethod
    def load_header_mem(cls, buf):
        a = np.frombuffer(buf, dtype=np.uint8)
        h = _P()
        _chk(load().pm_index_load_header_mem(a.ctypes.data, a.size, C.byref(h)))
        return cls(h)

    @classmethod
    def synth(cls, batch_id, n_docs, signature_size, num_hashes=1, term_size=31, seed=661,
              layout=PM_LAYOUT_AUTO, header_only=False):
        """measurement / test aid (libphylign_bench.so): a 661k-shaped synthetic index generated in HBM"""
        from . import bench_aids
        return cls(bench_aids.index_synth(batch_id, n_docs, signature_size, num_hashes, term_size, seed, layout, header_only))

    @classmethod
    def create(cls, names, signature_size, num_hashes=1, term_size=31, canonicalize=1, layout=PM_LAYOUT_AUTO, header_only=False):
        """an index made in place: header + names + a zeroed matrix in HBM (pm_index_create)"""
        blob = "".join(n + "\n" for n in names).encode()
        h = _P()
        _chk(load().pm_index_create(term_size, canonicalize, signature_size, num_hashes, blob, len(blob), len(names), layout,
                                    int(header_only), C.byref(h)))
        return cls(h)

    def matrix_device(self):
        """(device address, stride) of the resident matrix"""
        p, st = _P(), C.c_uint64()
        _chk(load().pm_index_matrix_device(self._h, C.byref(p), C.byref(st)))
        return p.value, st.value

    @classmethod
    def from_names(cls, names, term_size=31):
        blob = "".join(n + "\n" for n in names).encode()
        h = _P()
        _chk(load().pm_index_from_names(blob, len(blob), len(names), term_size, C.byref(h)))
        return cls(h)

    def names(self):
        return [self.doc_name(d) for d in range(self.info.n_docs)]

    def drop_matrix(self):
        _chk(load().pm_index_drop_matrix(self._h))

    def plant(self, rows, docs):
        """measurement / test aid (libphylign_bench.so): sets bit (rows[i], docs[i])"""
        from . import bench_aids
        bench_aids.index_plant(self._h, rows, docs)

    def plant_cluster(self, queries, q_first, q_step, seed=97):
        """measurement / test aid (libphylign_bench.so): makes this index the home batch of queries q_first,
        q_first + q_step, ... (see include/phylign_match_bench.h)"""
        from . import bench_aids
        bench_aids.index_plant_cluster(self, queries, q_first, q_step, seed)

    def read_rows(self, row0, n):
        out = np.zeros((n, self.info.row_bytes), dtype=np.uint8)
        _chk(load().pm_index_read_rows(self._h, row0, n, out.ctypes.data))
        return out

    def probe_gather(self, n_groups, lookups_per_group, **kw):
        """measurement aid (libphylign_bench.so): (ms, algorithmic bytes) of a pure random-row gather with k_scan's pattern"""
        from . import bench_aids
        return bench_aids.probe_gather(self, n_groups, lookups_per_group, **kw)

    @property
    def info(self):
        i = IndexInfo()
        _chk(load().pm_index_info(self._h, C.byref(i)))
        return i

    @property
    def device(self):
        """GPU ordinal that holds the matrix (-1: header-only handle)"""
        d = C.c_int()
        _chk(load().pm_index_device(self._h, C.byref(d)))
        return d.value

    def doc_name(self, d):
        n = C.c_size_t()
        p = load().pm_index_doc_name(self._h, d, C.byref(n))
        if not p:
            raise IndexError(d)
        return C.string_at(p, n.value).decode(errors="replace")

    def read_row(self, row):
        out = np.zeros(self.info.row_bytes, dtype=np.uint8)
        _chk(load().pm_index_read_row(self._h, row, out.ctypes.data))
        return out

    def free(self):
        if self._h:
            load().pm_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _buffer_ptr(buf):
    """(address, length, object to keep alive) of a bytes-like object, without copying it"""
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), _P).value if buf else None, len(buf), buf
    arr = np.frombuffer(buf, dtype=np.uint8)
    return (arr.ctypes.data if arr.size else None), arr.size, arr


def fasta_record_cuts(fasta, max_records):
    """offsets at which a prepared query file is cut into pieces of max_records records (see the header)"""
    ptr, n, _keep = _buffer_ptr(fasta)
    p, k = _P(), C.c_uint64()
    _chk(load().pm_fasta_record_cuts(ptr, n, max_records, C.byref(p), C.byref(k)))
    if not k.value:
        return []
    out = list((C.c_uint64 * k.value).from_address(p.value))
    load().pm_free(p)
    return out


def emit_merges_to(merges, path) -> int:
    """the 04_filter FASTA of a query file searched in chunks: the merges' records in chunk (= file) order"""
    total = 0
    for i, m in enumerate(merges):
        piece = 0 if len(merges) == 1 else (1 if i == 0 else (3 if i == len(merges) - 1 else 2))
        total += m.emit_to(path, piece)
    return total


class Queries:
    """A query FASTA parsed with the cobs CLI's record rules, resident in HBM."""

    def __init__(self, fasta: bytes, term_size=31, normalise=False):
        """normalise: `fasta` is an unprocessed FASTA/FASTQ; rules fix_query + concatenate_queries
        (Snakefile:314-352) are applied by the native parser"""
        h = _P()
        ptr, n, _keep = _buffer_ptr(fasta)                 # bytes or any buffer (a memoryview slice of a big file: no copy)
        _chk(load().pm_queries_parse_raw(ptr, n, term_size, int(bool(normalise)), C.byref(h)))
        self._h = h

    def fasta(self) -> bytes:
        """the prepared single-line FASTA this set stands for"""
        t, n = _P(), C.c_size_t()
        _chk(load().pm_queries_fasta(self._h, C.byref(t), C.byref(n)))
        out = C.string_at(t.value, n.value)
        load().pm_free(t)
        return out

    def count(self):
        nq, nt = C.c_uint64(), C.c_uint64()
        _chk(load().pm_queries_count(self._h, C.byref(nq), C.byref(nt)))
        return nq.value, nt.value

    def terms(self, i):
        n = C.c_uint64()
        _chk(load().pm_queries_terms(self._h, i, C.byref(n)))
        return n.value

    def hash_terms(self, canonicalize=1, num_hashes=1):
        _, nt = self.count()
        out = np.zeros(nt * num_hashes, dtype=np.uint64)
        _chk(load().pm_hash_terms(self._h, canonicalize, num_hashes, out.ctypes.data))
        return out

    def release_device(self):
        """frees the HBM copies (sequences, descriptors, hashes); the host side stays, the next search uploads again"""
        _chk(load().pm_queries_release_device(self._h))

    def device_bytes(self, num_hashes=1):
        """(HBM bytes held now, HBM bytes held while searched with num_hashes hash functions)"""
        a, b = C.c_uint64(), C.c_uint64()
        _chk(load().pm_queries_device_bytes(self._h, num_hashes, C.byref(a), C.byref(b)))
        return a.value, b.value

    def free(self):
        if self._h:
            for h in self.__dict__.pop("_aid_hashes", {}).values():      # device copies the measurement aids made
                h.free()
            load().pm_queries_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Result:
    def __init__(self, handle):
        self._h = handle

    @property
    def stats(self):
        s = Stats()
        _chk(load().pm_result_stats(self._h, C.byref(s)))
        return s

    def launches(self):
        """scan-kernel launches of this search: dicts with the kernel's template
        parameters, batches covered, algorithmic bytes and hipEvent ms"""
        n = C.c_size_t()
        _chk(load().pm_result_launches(self._h, None, 0, C.byref(n)))
        arr = (Launch * max(n.value, 1))()
        _chk(load().pm_result_launches(self._h, arr, n.value, C.byref(n)))
        return [{"kernel": f"k_scan<G={a.lanes_per_row or 'mixed'},P={a.planes},{'NH1' if a.num_hashes == 1 else 'NHn'}{',WQ' if a.wide_query else ''}>",
                 "n_batches": a.n_batches, "n_queries": a.n_queries,
                 "algorithmic_bytes": a.algorithmic_bytes, "ms": a.ms,
                 "fetched_bytes": a.fetched_bytes} for a in arr[: n.value]]

    def hits_device(self):
        p, n = _P(), C.c_uint64()
        _chk(load().pm_result_hits_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def copy_hits_device(self, dst_ptr, capacity, ordered=False):
        _chk(load().pm_result_copy_hits_device(self._h, dst_ptr, capacity, int(ordered)))

    def wait(self):
        """blocks until the GPU has finished this search (results of search_async)"""
        _chk(load().pm_result_wait(self._h))
        return self

    def ordered_device(self):
        """(device pointer, n) of the records after the device-side ordering (RCCL send buffer)"""
        p, n = _P(), C.c_uint64()
        _chk(load().pm_result_ordered_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def hits(self, copy=True):
        """numpy structured array (HIT_DTYPE) ordered (slot, query, score desc, doc asc);
        a count record (doc == PM_DOC_COUNT) leads the hits of a (query, slot) whose list
        was cut to the n best on the GPU.  copy=False returns a view of the library's pinned
        buffer: valid only until this Result is freed."""
        p, n = _P(), C.c_uint64()
        _chk(load().pm_result_hits_host(self._h, C.byref(p), C.byref(n)))
        if n.value == 0:
            return np.empty(0, dtype=HIT_DTYPE)
        buf = (C.c_char * (n.value * HIT_DTYPE.itemsize)).from_address(p.value)
        view = np.frombuffer(buf, dtype=HIT_DTYPE)
        return view.copy() if copy else view

    def slot_hits(self, slot):
        """the ordered records of ONE index of the search (position `slot` in the index list), read back on their own
        into a pooled pinned buffer: a Slice whose .hits view is valid until its free() (or `with`)"""
        p, n, tok = _P(), C.c_uint64(), _P()
        _chk(load().pm_result_slot_hits(self._h, slot, C.byref(p), C.byref(n), C.byref(tok)))
        return Slice(p.value, n.value, tok)

    def free(self):
        if self._h:
            load().pm_result_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Slice:
    """records of one index of a search on the host (Result.slot_hits)"""

    def __init__(self, ptr, n, tok):
        self._tok = tok
        if n:
            buf = (C.c_char * (n * HIT_DTYPE.itemsize)).from_address(ptr)
            self.hits = np.frombuffer(buf, dtype=HIT_DTYPE)
        else:
            self.hits = np.empty(0, dtype=HIT_DTYPE)

    def free(self):
        if self._tok:
            self.hits = np.empty(0, dtype=HIT_DTYPE)
            load().pm_slice_free(self._tok)
            self._tok = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def search(indexes, queries: Queries, threshold: float, slot_base=0, nb_best_hits=0, parts=None) -> Result:
    """nb_best_hits > 0: per (query, index) keep the n best documents + ties (on the GPU)"""
    if parts is not None:
        res = search_async(indexes, queries, threshold, slot_base, nb_best_hits, parts)
        res.wait()
        return res
    arr = (_P * len(indexes))(*[ix._h for ix in indexes])
    h = _P()
    _chk(load().pm_search(arr, len(indexes), queries._h, threshold, nb_best_hits, slot_base, C.byref(h)))
    return Result(h)


def search_async(indexes, queries: Queries, threshold: float, slot_base=0, nb_best_hits=0, parts=None) -> Result:
    """search() that returns once the kernels are queued; Result.wait() / any getter blocks.
    parts: per index None (all queries) or (lo, hi, den): the share of the queries this process searches the index
    with (pm_search_async_parts: a batch resident on several ranks)"""
    arr = (_P * len(indexes))(*[ix._h for ix in indexes])
    h = _P()
    if parts is None or all(p is None for p in parts):
        _chk(load().pm_search_async(arr, len(indexes), queries._h, threshold, nb_best_hits, slot_base, C.byref(h)))
        return Result(h)
    assert len(parts) == len(indexes)
    pa = np.zeros((len(indexes), 3), dtype=np.uint32)
    for i, p_ in enumerate(parts):
        if p_ is not None:
            pa[i] = p_
    _chk(load().pm_search_async_parts(arr, len(indexes), queries._h, threshold, nb_best_hits, slot_base, pa.ctypes.data, C.byref(h)))
    return Result(h)


class Merge:
    """04_filter state (scripts/filter_queries.py): best `keep` matches (+ties) per query across batches"""

    def __init__(self, queries: Queries, keep=100):
        h = _P()
        _chk(load().pm_merge_create(queries._h, keep, C.byref(h)))
        self._h, self._q = h, [queries]

    def extend(self, queries: Queries):
        """the next piece of the same query file (its records continue the numbering; names are one namespace)"""
        _chk(load().pm_merge_extend(self._h, queries._h))
        self._q.append(queries)

    def add(self, batch: str, index: Index, hits, slot=0, nb_best_hits=-1, piece=0):
        """piece: the records' query numbers count inside that piece of the query file; -1: through the whole file"""
        hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
        _chk(load().pm_merge_add_piece(self._h, piece, batch.encode(), index._h, hits.ctypes.data, hits.size, slot, nb_best_hits))

    def batches(self):
        """batch names in the order of their numbers (the `slot` of export()'s records)"""
        t, n = _P(), C.c_size_t()
        _chk(load().pm_merge_batches(self._h, C.byref(t), C.byref(n)))
        out = C.string_at(t.value, n.value).decode()
        load().pm_free(t)
        return out.split("\n")[:-1] if out else []

    def add_text(self, batch: str, text: bytes):
        """the 03_match text of one batch (after gunzip), as scripts/filter_queries.py reads it"""
        _chk(load().pm_merge_add_text(self._h, batch.encode(), text, len(text)))

    def export(self):
        """what is kept so far as a HIT_DTYPE array (query numbered through the whole file, slot = number of the batch:
        batches())"""
        p, n = _P(), C.c_uint64()
        _chk(load().pm_merge_export(self._h, C.byref(p), C.byref(n)))
        buf = (C.c_char * (n.value * HIT_DTYPE.itemsize)).from_address(p.value) if n.value else b""
        out = np.frombuffer(buf, dtype=HIT_DTYPE).copy()
        load().pm_free(p)
        return out

    def emit_to(self, path, piece=0) -> int:
        """writes the 04_filter FASTA to `path` (atomically); returns its size.  piece: 0 = the whole file; for a query
        file searched in chunks 1 = first merge, 2 = a middle one, 3 = the last (the file appears with it)"""
        n = C.c_uint64()
        _chk(load().pm_merge_emit_file_piece(self._h, os.fsencode(path), piece, C.byref(n)))
        return n.value

    def emit(self) -> bytes:
        t, n = _P(), C.c_size_t()
        _chk(load().pm_merge_emit(self._h, C.byref(t), C.byref(n)))
        out = C.string_at(t.value, n.value)
        load().pm_free(t)
        return out

    def free(self):
        if self._h:
            load().pm_merge_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sort_hits(hits):
    """orders a HIT_DTYPE array in place the way cobs orders result lines"""
    assert hits.dtype == HIT_DTYPE and hits.flags.c_contiguous
    load().pm_hits_sort(hits.ctypes.data, hits.size)
    return hits


def format_hits(index: Index, queries: Queries, hits, slot=0, nb_best_hits=-1) -> bytes:
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    t, n = _P(), C.c_size_t()
    _chk(load().pm_format_hits(index._h, queries._h, hits.ctypes.data, hits.size, slot, nb_best_hits,
                               C.byref(t), C.byref(n)))
    out = C.string_at(t.value, n.value)
    load().pm_free(t)
    return out


def format_hits_gz(index: Index, queries: Queries, hits, path, slot=0, nb_best_hits=-1, level=1, piece=0):
    """writes the batch's 03_match file (text of format_hits, gzip members, atomic rename); returns (text bytes, gz bytes).
    piece: 0 = the whole file; for a query set searched in chunks 1 = first piece, 2 = middle, 3 = last (see the header)"""
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    t, z = C.c_uint64(), C.c_uint64()
    _chk(load().pm_format_hits_gz_piece(index._h, queries._h, hits.ctypes.data, hits.size, slot, nb_best_hits,
                                        os.fsencode(path), level, piece, C.byref(t), C.byref(z)))
    return t.value, z.value


def gzip_fast(text: bytes) -> bytes:
    """`gzip --fast` of a result text by the library's own encoder (gzip members of ~1 MiB, built in parallel)"""
    g, n = _P(), C.c_size_t()
    _chk(load().pm_gzip_fast(text, len(text), C.byref(g), C.byref(n)))
    out = C.string_at(g.value, n.value)
    load().pm_free(g)
    return out


def format_hits_limit(index: Index, queries: Queries, hits, slot=0, limit=0) -> bytes:
    """plain cobs text with at most `limit` result lines per query (`cobs query -l`)"""
    hits = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    t, n = _P(), C.c_size_t()
    _chk(load().pm_format_hits_limit(index._h, queries._h, hits.ctypes.data, hits.size, slot, limit, C.byref(t), C.byref(n)))
    out = C.string_at(t.value, n.value)
    load().pm_free(t)
    return out


def query_text(index: Index, fasta: bytes, threshold: float, nb_best_hits=-1) -> bytes:
    t, n = _P(), C.c_size_t()
    _chk(load().pm_query_text(index._h, fasta, len(fasta), threshold, nb_best_hits, C.byref(t), C.byref(n)))
    out = C.string_at(t.value, n.value)
    load().pm_free(t)
    return out
