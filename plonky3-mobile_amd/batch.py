"""Batches of independent fib_air proofs across the GPUs of one node (BASELINE configs[3]).

Proofs are independent, so there is no collective on the data path: instance i goes to rank i mod world.
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only to
scatter the instance descriptors (a, b) from rank 0 and to gather the proof bytes back."""
import numpy as np
import torch
import torch.distributed as dist


def shard_instances(n_total, rank, world):
    """Indices of the instances rank `rank` proves: i = rank, rank + world, ..."""
    return list(range(rank, n_total, world))


def scatter_descriptors(instances, device="cpu"):
    """Rank 0 holds `instances` = list of (a, b); every rank receives its shard (list of (index, a, b))."""
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([len(instances) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, 0)
    n = int(n.item())
    per = (n + world - 1) // world
    recv = torch.full((per, 3), -1, dtype=torch.int64, device=device)
    if rank == 0:
        chunks = []
        for r in range(world):
            rows = [(i, instances[i][0], instances[i][1]) for i in shard_instances(n, r, world)]
            rows += [(-1, 0, 0)] * (per - len(rows))
            chunks.append(torch.tensor(rows, dtype=torch.int64, device=device).reshape(per, 3))
        dist.scatter(recv, chunks, src=0)
    else:
        dist.scatter(recv, None, src=0)
    return [(int(i), int(a), int(b)) for i, a, b in recv.cpu().tolist() if i >= 0]


def gather_proofs(local, n_total, device="cpu"):
    """local: list of (index, proof bytes) of this rank.  Returns on rank 0 the list of all n_total proofs in
    instance order (None elsewhere).  Proofs are padded to the longest one for a fixed-size all_gather."""
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_total + world - 1) // world
    max_len = torch.tensor([max([len(p) for _, p in local], default=0)], dtype=torch.int64, device=device)
    dist.all_reduce(max_len, op=dist.ReduceOp.MAX)
    width = int(max_len.item())
    buf = torch.zeros((per, width), dtype=torch.uint8, device=device)
    meta = torch.full((per, 2), -1, dtype=torch.int64, device=device)  # (instance index, length)
    for k, (i, p) in enumerate(local):
        buf[k, : len(p)] = torch.from_numpy(np.frombuffer(p, dtype=np.uint8).copy()).to(device)
        meta[k, 0], meta[k, 1] = i, len(p)
    bufs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    metas = [torch.empty_like(meta) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, bufs, dst=0)
    dist.gather(meta, metas, dst=0)
    if rank != 0:
        return None
    out = [None] * n_total
    for b, m in zip(bufs, metas):
        b, m = b.cpu().numpy(), m.cpu().tolist()
        for k, (i, ln) in enumerate(m):
            if i >= 0:
                out[i] = b[k, :ln].tobytes()
    return out
