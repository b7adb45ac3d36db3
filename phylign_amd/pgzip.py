"""Parallel `gzip --fast` for the 03_match files (Snakefile:468 pipes the post-filtered
text through `gzip --fast`).  The text is cut at line boundaries into chunks that are
deflated in parallel (level 1: the library's own encoder, pm_gzip_fast; other levels: zlib
on a thread pool, which releases the GIL) and written as consecutive gzip MEMBERS: a multi-member file is a valid gzip stream -- `gzip -dc`, Python's gzip /
xopen (scripts/filter_queries.py:27-66 reads through xopen) decode it to the same
bytes -- so the consumer side is unchanged."""
import os
import zlib
from concurrent.futures import ThreadPoolExecutor

CHUNK = 4 << 20


def _member(data, level):
    c = zlib.compressobj(level, zlib.DEFLATED, 31)       # wbits 31: gzip container
    return c.compress(data) + c.flush()


def split_lines(text, chunk=CHUNK):
    """cuts `text` into pieces of about `chunk` bytes that end on a newline"""
    out, p, n = [], 0, len(text)
    while p < n:
        e = min(n, p + chunk)
        if e < n:
            nl = text.find(b"\n", e)
            e = n if nl < 0 else nl + 1
        out.append(text[p:e])
        p = e
    return out


def compress(text, level=1, threads=None, pool=None):
    """gzip bytes of `text` (level 1 = `gzip --fast`); several members when the text is long"""
    if level == 1:
        from . import _lib as pm
        return pm.gzip_fast(bytes(text))
    parts = split_lines(text) or [b""]
    if len(parts) == 1:
        return _member(parts[0], level)
    if pool is not None:
        return b"".join(pool.map(lambda d: _member(d, level), parts))
    with ThreadPoolExecutor(max_workers=threads or min(16, os.cpu_count() or 1)) as ex:
        return b"".join(ex.map(lambda d: _member(d, level), parts))


def write(path, text, level=1, pool=None):
    """atomic: never leaves a partial file that looks complete"""
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(compress(text, level, pool=pool))
    os.replace(tmp, path)
