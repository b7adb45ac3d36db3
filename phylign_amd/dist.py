"""Multi-GPU leg of the matching stage: batches are sharded
statically over ranks (workload.assign_batches); the only exchange is one
gather of hit records to rank 0 at the end -- counts by all_gather, payload by
point-to-point send/recv (RCCL over xGMI with the `nccl` backend: every peer
has its own link into the root, so the gather is link-parallel).  Works
unchanged on `gloo` (CPU tensors) for the world_size-2 tests."""
import torch
import torch.distributed as dist


def gather_hits(local, dst=0, group=None):
    """local: int32 tensor [n, 4] of pm_hit_t records (device tensor with nccl,
    CPU tensor with gloo).  Returns the concatenation over ranks in rank order
    on `dst`, None elsewhere."""
    if not dist.is_initialized():
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert local.dtype == torch.int32 and local.dim() == 2 and local.shape[1] == 4
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    if rank == dst:
        parts, ops = [], []
        for r in range(world):
            if r == dst:
                parts.append(local)
                continue
            buf = torch.empty((counts[r], 4), dtype=torch.int32, device=local.device)
            parts.append(buf)
            if counts[r]:
                ops.append(dist.P2POp(dist.irecv, buf, r, group=group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return torch.cat(parts, dim=0)
    if local.shape[0]:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), dst, group=group)]):
            w.wait()
    return None


class PackedGather:
    """One-collective form of gather_hits for a step loop: every rank owns fixed
    [cap + 1, 4] int32 buffers whose row 0 carries the record count; one
    `dist.gather` moves all of them to rank 0 (with RCCL: 7 peer-to-root
    transfers over 7 distinct xGMI links).  A rank with more than `cap` records
    announces its count and ships the records in a second, point-to-point
    message, so the result never depends on `cap`.

    The send buffers alternate (`depth` of them): with RCCL the collective is
    only QUEUED when dist.gather returns, and the next step fills the OTHER
    buffer, so a non-root rank never waits for its send -- an event recorded
    behind the collective is waited for only when the buffer comes round again,
    one whole step later.  The gather of step i thus overlaps the scan of step
    i + 1 on every rank."""

    def __init__(self, cap, device, group=None, depth=2):
        self.cap, self.group, self.device = int(cap), group, device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.packs = [torch.zeros((self.cap + 1, 4), dtype=torch.int32, device=device) for _ in range(max(1, depth))]
        self.sent = [None] * len(self.packs)         # event behind the last collective that read packs[i]
        self.turn = 0
        self.active = dist.is_initialized()          # a process group of ONE rank still goes through the collective
        self.recv = [torch.zeros_like(self.packs[0]) for _ in range(self.world)] if self.rank == 0 and self.active else None
        self.waited_s = 0.0                          # host time spent waiting for a buffer to come free (should stay ~0)

    @property
    def pack(self):
        return self.packs[self.turn]

    def records_view(self):
        """rows 1.. of the current send buffer: the place to copy up to `cap` records into (waits until the collective
        that last read this buffer has finished)"""
        ev = self.sent[self.turn]
        if ev is not None:
            if not ev.query():
                import time
                t0 = time.perf_counter()
                ev.synchronize()
                self.waited_s += time.perf_counter() - t0
            self.sent[self.turn] = None
        return self.packs[self.turn][1:]

    def gather(self, count, overflow=None, dst=0):
        """count: number of valid rows in records_view(), or the true count when
        it exceeds cap and `overflow` (int32 [count, 4]) holds all records."""
        assert dst == 0
        pack = self.packs[self.turn]
        if not self.active:
            return overflow if count > self.cap else pack[1:1 + count]
        self.records_view()                           # (no-op when the caller filled the buffer through records_view())
        pack[0, 0] = int(count)
        dist.gather(pack, self.recv, dst=0, group=self.group)
        if self.rank != 0:
            if count > self.cap:
                dist.send(overflow.contiguous(), 0, group=self.group)
                if pack.is_cuda:                      # `overflow` is the caller's: it must have left before we return
                    torch.cuda.current_stream(pack.device).synchronize()
            elif pack.is_cuda:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(pack.device))      # the stream waits for the collective (work.wait())
                self.sent[self.turn] = ev
            self.turn = (self.turn + 1) % len(self.packs)
            return None
        self.turn = (self.turn + 1) % len(self.packs)
        counts = torch.stack([r[0, 0] for r in self.recv]).cpu().tolist()
        parts = []
        for r, c in enumerate(counts):
            if c <= self.cap:
                parts.append(self.recv[r][1:1 + c])
            elif r == 0:
                parts.append(overflow)
            else:
                buf = torch.empty((c, 4), dtype=torch.int32, device=self.device)
                dist.recv(buf, r, group=self.group)
                parts.append(buf)
        return torch.cat(parts, dim=0)
