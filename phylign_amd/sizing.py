"""RAM / thread sizing of one 03_match job: behavioural mirror of the helper
functions the reference Snakefile uses for the COBS rules (Snakefile:41-121),
pinned by tests/golden/sizing.tsv.  On the MI355X path they size the host side
of match_stage -- how many xz decoders run at once and how much host RAM they
may take together (stage_plan / HostRam below): the index itself lives in HBM,
so what the reference reserves per job as `max_ram_mb` (index MB + decoder MB,
Snakefile:440-449) is here the decoder's share only."""
import re
import threading

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


# ---- the stage's use of the rules above -------------------------------------------------------------------------------
STAGING_MB = 64          # two pinned 32 MiB chunks per loader (pm_index.cpp stream_matrix)


def loader_host_mb(batch, sizes_path):
    """host RAM one loader of `batch` takes while it decodes: the xz decoder (Snakefile:64-69) + the pinned staging.
    The index bytes the reference adds on top (Snakefile:72-82) are in HBM here."""
    try:
        return xz_ram_mb(batch, sizes_path) + STAGING_MB
    except (KeyError, OSError, TypeError):
        return 1536 + STAGING_MB                       # not in the table: the 661k files' decoder size (1.5 GiB)


def stage_plan(batches, sizes_path, cpus, max_ram_gb, loaders=0):
    """(number of loader threads, host-RAM budget in MB, {batch: MB one loader of it needs}).  The budget is the
    reference's `max_ram_gb` (config.yaml:53-58: the RAM its jobs may hold together; Snakemake admits a job while the sum
    of the running jobs' max_ram_mb fits).  loaders = 0: as many as the CPUs allow (all but 4, at least 4, at most 16: one
    xz stream decodes 0.1-0.2 GB/s on one core) and as fit the budget at the largest decoder size."""
    need = {b: loader_host_mb(b, sizes_path) for b in batches}
    budget_mb = int(max_ram_gb * 1024)
    if loaders <= 0:
        loaders = max(4, min(16, cpus - 4))
        if need and budget_mb > 0:
            loaders = max(1, min(loaders, budget_mb // max(need.values())))
    return loaders, budget_mb, need


class HostRam:
    """admission of loaders to the host-RAM budget, like Snakemake's max_ram_mb resource: a loader waits while the RAM the
    running ones hold plus its own exceeds the budget (one that needs more than the whole budget runs alone)"""

    def __init__(self, budget_mb):
        self.budget, self.held, self.peak = budget_mb, 0, 0
        self.cv = threading.Condition()

    def acquire(self, mb):
        with self.cv:
            while self.budget > 0 and self.held > 0 and self.held + mb > self.budget:
                self.cv.wait()
            self.held += mb
            self.peak = max(self.peak, self.held)

    def release(self, mb):
        with self.cv:
            self.held -= mb
            self.cv.notify_all()
