"""RAM / thread sizing of one 03_match job: behavioural mirror of the helper
functions the reference Snakefile uses for the COBS rules (Snakefile:41-121),
pinned by tests/golden/sizing.tsv.  On the MI355X path they size the host side
(xz decoder RAM, pinned staging) -- the index itself lives in HBM."""
import re

MIB = 1024 * 1024


def index_metadata(batch, sizes_path):
    """(uncompressed bytes, xz decoder RAM bytes) of `batch` from the
    decompressed_indexes_sizes table (Snakefile:41-57)."""
    with open(sizes_path) as f:
        for line in f:
            path, nbytes, xz_ram = line.split()
            if path.rsplit("/", 1)[-1].replace(".cobs_classic.xz", "") == batch:
                return int(nbytes), int(xz_ram)
    raise KeyError(f"batch {batch} not found in {sizes_path}")


def xz_ram_mb(batch, sizes_path):
    return int(index_metadata(batch, sizes_path)[1] / MIB) + 1           # Snakefile:64-69


def batch_ram_mb(batch, sizes_path, ignore_ram=False, streaming=True):
    """Snakefile:72-82: index MB (+ xz decoder MB when streaming)."""
    if ignore_ram:
        return 0
    extra = xz_ram_mb(batch, sizes_path) if streaming else 0
    return int(index_metadata(batch, sizes_path)[0] / MIB) + 1 + extra


def cobs_threads(batch, sizes_path, cobs_threads_cfg, cores, max_ram_gb, streaming=True):
    """Snakefile:93-121: explicit number, 'auto' (= all cores) or 'auto(N)'."""
    cfg = str(cobs_threads_cfg)
    if not cfg.startswith("auto"):
        return int(cfg)
    if cfg == "auto":
        cap = cores
    else:
        m = re.findall(r"auto\((\d+)\)", cfg)
        if len(m) != 1:
            raise ValueError("cannot parse cobs_threads")
        cap = int(m[0])
    share = batch_ram_mb(batch, sizes_path, False, streaming) / (int(max_ram_gb) * 1024)
    n = min(max(round(share * cap), 1), cap)
    return cap if n > cap / 2 else n
