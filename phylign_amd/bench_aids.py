"""ctypes binding of libphylign_bench.so (include/phylign_match_bench.h): measurement and test aids --
661k-shaped synthetic indexes generated in HBM, planted hits, "home batch" clusters, the gather probe.
Loaded by bench.py, tools/ and tests/ (through Index.synth / .plant / .plant_cluster / .probe_gather);
the drop-in path never imports this module."""
import ctypes as C
import os

import numpy as np

from . import _lib as pm

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PHYLIGN_BENCH_LIB") or os.path.join(_HERE, "libphylign_bench.so")
_P = C.c_void_p

SYMBOLS = [
    ("pm_bench_last_error", C.c_char_p, []),
    ("pm_bench_index_synth", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.POINTER(_P)]),
    ("pm_bench_index_plant", C.c_int, [_P, _P, _P, C.c_size_t]),
    ("pm_bench_hashes_create", C.c_int, [_P, C.c_int, C.c_uint32, C.POINTER(_P)]),
    ("pm_bench_hashes_free", None, [_P]),
    ("pm_bench_index_plant_cluster", C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint64]),
    ("pm_bench_index_save", C.c_int, [_P, C.c_char_p]),
    ("pm_bench_unique_rows", C.c_int, [_P, _P, C.POINTER(C.c_uint64)]),
    ("pm_bench_index_correlate", C.c_int, [_P, C.c_uint64, C.c_uint32]),
    ("pm_bench_probe_gather", C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
]
_aids = None


def load():
    """dlopen the aids library (after the product library it links against); raises when it was not built"""
    global _aids
    if _aids is None:
        pm.load()
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: python -m phylign_amd.build builds it next to libphylign_match.so")
        # the product library must be visible to the loader under its SONAME first
        C.CDLL(pm.LIB_PATH, mode=C.RTLD_GLOBAL)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _aids = L
    return _aids


def _chk(rc):
    if rc != 0:
        raise pm.PMError(rc, load().pm_bench_last_error().decode(errors="replace"))


def index_synth(batch_id, n_docs, signature_size, num_hashes=1, term_size=31, seed=661, layout=0, header_only=False):
    h = _P()
    _chk(load().pm_bench_index_synth(batch_id, n_docs, signature_size, num_hashes, term_size, seed, layout, int(header_only), C.byref(h)))
    return h


def index_plant(index_handle, rows, docs):
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    docs = np.ascontiguousarray(docs, dtype=np.uint32)
    assert rows.size == docs.size
    _chk(load().pm_bench_index_plant(index_handle, rows.ctypes.data, docs.ctypes.data, rows.size))


class Hashes:
    """device copy of a query set's hashes in the aids' layout, shared by the plantings of many batches"""

    def __init__(self, queries, canonicalize, num_hashes):
        h = _P()
        _chk(load().pm_bench_hashes_create(queries._h, canonicalize, num_hashes, C.byref(h)))
        self._h = h

    def free(self):
        if self._h:
            load().pm_bench_hashes_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _hashes_of(index, queries):
    info = index.info
    key = (int(info.canonicalize), int(info.num_hashes))
    cache = queries.__dict__.setdefault("_aid_hashes", {})
    if key not in cache:
        cache[key] = Hashes(queries, *key)
    return cache[key]


def index_plant_cluster(index, queries, q_first, q_step, seed=97):
    _chk(load().pm_bench_index_plant_cluster(index._h, _hashes_of(index, queries)._h, q_first, q_step, seed))


def unique_rows(index, queries):
    """distinct signature rows of `index` that the k-mers of `queries` map to (many-queries regime)"""
    n = C.c_uint64()
    _chk(load().pm_bench_unique_rows(index._h, _hashes_of(index, queries)._h, C.byref(n)))
    return n.value


def probe_gather(index, n_groups, lookups_per_group, mode=None, flavor=None, unroll=None):
    """(ms, algorithmic bytes) of a pure random-row gather with k_scan's access pattern; mode / flavor / unroll default to
    the PM_PROBE_MODE / PM_PROBE_FLAVOR / PM_PROBE_UNROLL variables the calibration scripts under tools/ set"""
    mode = int(os.environ.get("PM_PROBE_MODE", "0")) if mode is None else mode
    flavor = int(os.environ.get("PM_PROBE_FLAVOR", "0")) if flavor is None else flavor
    unroll = int(os.environ.get("PM_PROBE_UNROLL", "8")) if unroll is None else unroll
    ms, nb = C.c_double(), C.c_uint64()
    _chk(load().pm_bench_probe_gather(index._h, n_groups, lookups_per_group, mode, flavor, unroll, C.byref(ms), C.byref(nb)))
    return ms.value, nb.value


def index_correlate(index, seed=7, flip_log2=7):
    """cold-path timings only: overwrites the resident matrix with compressible content (see the header)"""
    _chk(load().pm_bench_index_correlate(index._h, seed, flip_log2))


def index_save(index, path):
    """writes a resident classic index back as a .cobs_classic file"""
    _chk(load().pm_bench_index_save(index._h, os.fsencode(path)))
