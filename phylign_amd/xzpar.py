"""Decoding ONE multi-block .xz file on several threads.

The reference streams every index through a single-threaded `xzcat`
(scripts/run_cobs_streaming.sh:27); the stage runs one such decoder per loader
thread, which keeps all CPUs busy while a rank has at least as many compressed
batches as CPUs.  With FEWER batches than CPUs (data/batches_small.txt: three)
the decode of the largest file is the whole cold time, and a file that was
written by `xz -T` consists of independent blocks that can be decoded side by
side.  The image's xz 5.2.5 has no threaded decoder and liblzma's headers are
absent, so this module reads the container itself (stream footer -> index ->
block offsets, per block: header -> LZMA2 dictionary size) and hands every
block's raw LZMA2 data to Python's `lzma` module, which releases the GIL while
it decodes.  Anything it does not understand -- several streams, other filters
than one LZMA2 (checked on EVERY block header, not only the index), a single
block, a path that is not a regular file -- makes `plan()` return None and the
caller falls back to `xzcat`.

Container layout: the .xz file format specification 1.0.4, sections 2.1 (stream
header / footer), 3.1 (block header), 4 (index)."""
import lzma
import os
import stat
import struct
import threading
import zlib
from collections import namedtuple
from concurrent.futures import ThreadPoolExecutor

Block = namedtuple("Block", "offset unpadded_size uncompressed_size")
Plan = namedtuple("Plan", "path blocks check_size uncompressed_size")
_HEADER_MAGIC = b"\xfd7zXZ\x00"
_FOOTER_MAGIC = b"YZ"
_CHECK_SIZE = {0: 0, 1: 4, 4: 8, 10: 32}        # none, CRC32, CRC64, SHA-256


def _varint(buf, pos):
    """(value, next position) of the multibyte integer at buf[pos:] (spec 1.2); raises ValueError when malformed"""
    val, shift = 0, 0
    for i in range(9):
        if pos + i >= len(buf):
            raise ValueError("truncated multibyte integer")
        b = buf[pos + i]
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            if b == 0 and i > 0:
                raise ValueError("non-minimal multibyte integer")
            return val, pos + i + 1
        shift += 7
    raise ValueError("multibyte integer longer than 9 bytes")


def plan(path, min_blocks=2):
    """Block table of a single-stream .xz file whose blocks can be decoded independently, or None (then use xzcat)."""
    try:
        st = os.stat(path)
        if not stat.S_ISREG(st.st_mode):
            return None                                      # a pipe / device: never opened here (reading would eat its header)
        size = st.st_size
        if size < 32:
            return None
        with open(path, "rb") as f:
            head = f.read(12)
            if size < 32 or head[:6] != _HEADER_MAGIC or zlib.crc32(head[6:8]) != struct.unpack("<I", head[8:12])[0]:
                return None
            f.seek(size - 12)
            foot = f.read(12)
            if foot[10:12] != _FOOTER_MAGIC or foot[8:10] != head[6:8]:
                return None                                  # stream padding / another stream behind: not handled here
            if zlib.crc32(foot[4:10]) != struct.unpack("<I", foot[0:4])[0]:
                return None
            check_size = _CHECK_SIZE.get(head[7] & 0x0F)
            if head[6] != 0 or check_size is None:
                return None
            index_size = (struct.unpack("<I", foot[4:8])[0] + 1) * 4
            if index_size + 24 > size:
                return None
            f.seek(size - 12 - index_size)
            idx = f.read(index_size)
        if idx[0] != 0 or zlib.crc32(idx[:-4]) != struct.unpack("<I", idx[-4:])[0]:
            return None
        n, pos = _varint(idx, 1)
        if n < min_blocks or n > (1 << 24):
            return None
        blocks, off, total = [], 12, 0
        for _ in range(n):
            unpadded, pos = _varint(idx, pos)
            uncomp, pos = _varint(idx, pos)
            if unpadded < 5 or off + unpadded > size - 12 - index_size:
                return None
            blocks.append(Block(off, unpadded, uncomp))
            off += (unpadded + 3) & ~3
            total += uncomp
        if off != size - 12 - index_size:
            return None                                      # blocks + index + footer must be the whole file (one stream)
        # every block must be what _decode_block handles (one LZMA2 filter): a file written with, say, --delta or --x86 in
        # front of LZMA2 has a perfectly good index, and finding that out in a worker would abort a load xzcat can do
        fd = os.open(path, os.O_RDONLY)
        try:
            for blk in blocks:
                _block_header(_pread_all(fd, min(1024, blk.unpadded_size), blk.offset), blk.unpadded_size)
        finally:
            os.close(fd)
        return Plan(path, blocks, check_size, total)
    except (OSError, ValueError, IndexError, struct.error):
        return None


def _pread_all(fd, n, offset):
    """n bytes at offset; one os.pread returns at most 2 GiB - 4 KiB on Linux, a block may be larger"""
    parts, got = [], 0
    while got < n:
        part = os.pread(fd, min(n - got, 1 << 30), offset + got)
        if not part:
            raise ValueError("short read: the file ends inside a block")
        parts.append(part)
        got += len(part)
    return parts[0] if len(parts) == 1 else b"".join(parts)


def _block_header(raw, unpadded_size):
    """(header size, LZMA2 dictionary size) of the block header at raw[0:] (spec 3.1); ValueError for anything but ONE
    LZMA2 filter with its one properties byte"""
    if not raw or raw[0] == 0:
        raise ValueError("index indicator where a block header should be")
    hsize = (raw[0] + 1) * 4
    if hsize > len(raw) or hsize > unpadded_size or zlib.crc32(raw[:hsize - 4]) != struct.unpack("<I", raw[hsize - 4:hsize])[0]:
        raise ValueError("block header CRC mismatch")
    flags = raw[1]
    if flags & 0x3C or (flags & 3) != 0:
        raise ValueError("block uses a filter chain this decoder does not handle")
    pos = 2
    if flags & 0x40:
        _, pos = _varint(raw, pos)
    if flags & 0x80:
        _, pos = _varint(raw, pos)
    fid, pos = _varint(raw, pos)
    psize, pos = _varint(raw, pos)
    if fid != 0x21 or psize != 1 or pos >= hsize - 4:
        raise ValueError("block is not plain LZMA2")
    bits = raw[pos] & 0x3F
    if bits > 40:
        raise ValueError("bad LZMA2 dictionary size")
    return hsize, (0xFFFFFFFF if bits == 40 else (2 | (bits & 1)) << (bits // 2 + 11))


def _decode_block(fd, blk, check_size):
    raw = _pread_all(fd, blk.unpadded_size, blk.offset)
    hsize, dict_size = _block_header(raw, blk.unpadded_size)
    dec = lzma.LZMADecompressor(format=lzma.FORMAT_RAW, filters=[{"id": lzma.FILTER_LZMA2, "dict_size": dict_size}])
    out = dec.decompress(raw[hsize:len(raw) - check_size])
    if len(out) != blk.uncompressed_size:
        raise ValueError(f"block decoded to {len(out)} bytes, the index says {blk.uncompressed_size}")
    return out


class ParallelXz:
    """`xzcat` of one file on `threads` threads: .stdout is the read end of a pipe (a file object), .wait() returns 0 when
    the whole file went through and 1 otherwise -- the two things the stage uses of a Popen.  Blocks are decoded at most
    `threads + 1` ahead of the writer, so memory stays at a few blocks."""

    def __init__(self, pl, threads):
        self.plan, self.threads = pl, max(1, int(threads))
        r, w = os.pipe()
        self.stdout = os.fdopen(r, "rb", buffering=0)
        self._w = w
        self.returncode = None
        self.error = None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        fd = -1
        try:
            fd = os.open(self.plan.path, os.O_RDONLY)
            with ThreadPoolExecutor(max_workers=self.threads) as pool:
                window, nxt = [], 0
                blocks = self.plan.blocks
                while nxt < len(blocks) or window:
                    while nxt < len(blocks) and len(window) <= self.threads:
                        window.append(pool.submit(_decode_block, fd, blocks[nxt], self.plan.check_size))
                        nxt += 1
                    data = window.pop(0).result()
                    view = memoryview(data)
                    while view:
                        view = view[os.write(self._w, view[:1 << 20]):]
            self.returncode = 0
        except BaseException as e:                              # noqa: BLE001 -- reported through wait(), like a decoder's exit status
            self.error, self.returncode = e, 1
        finally:
            if fd >= 0:
                os.close(fd)
            os.close(self._w)                                   # EOF for the reader (a short stream when something failed)

    def wait(self):
        self._t.join()
        return self.returncode
