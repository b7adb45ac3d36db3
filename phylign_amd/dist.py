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
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
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
