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


class _PendingGather:
    """An in-flight gather of one step's proofs (two async collectives); wait() returns the proofs on rank 0."""

    def __init__(self, works, bufs, metas, n_total, keep):
        self.works, self.bufs, self.metas, self.n_total, self.keep = works, bufs, metas, n_total, keep

    def wait(self):
        for w in self.works:
            w.wait()
        if self.bufs is None:
            return None
        out = [None] * self.n_total
        for b, m in zip(self.bufs, self.metas):
            b, m = b.cpu().numpy(), m.cpu().tolist()
            for k, (i, ln) in enumerate(m):
                if i >= 0:
                    out[i] = b[k, :ln].tobytes()
        return out


_STAGING = {}


def _staging(per, width, device):
    """Two alternating (pinned host, device) staging pairs per shape: the previous step's gather may still be
    reading one while the next step fills the other."""
    key = (per, width, str(device))
    ent = _STAGING.get(key)
    if ent is None:
        pairs = []
        for _ in range(2):
            h = torch.zeros((per, width), dtype=torch.uint8)
            if str(device) != "cpu":
                h = h.pin_memory()
                pairs.append((h, torch.empty((per, width), dtype=torch.uint8, device=device)))
            else:
                pairs.append((h, h))
        ent = _STAGING[key] = {"pairs": pairs, "turn": 0}
    ent["turn"] ^= 1
    return ent["pairs"][ent["turn"]]


def gather_proofs_async(local, n_total, device="cpu"):
    """local: list of (index, proof bytes) of this rank.  Starts the gather to rank 0 and returns a handle whose
    wait() gives, on rank 0, all n_total proofs in instance order (None elsewhere).  Proofs are staged in one host
    buffer padded to the longest one (a single H2D copy), so the collective has a fixed shape; being asynchronous
    it overlaps the next step's proving."""
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_total + world - 1) // world
    max_len = torch.tensor([max([len(p) for _, p in local], default=0)], dtype=torch.int64, device=device)
    dist.all_reduce(max_len, op=dist.ReduceOp.MAX)
    width = int(max_len.item())
    hbuf, buf = _staging(per, width, device)
    host = hbuf.numpy()
    hmeta = np.full((per, 2), -1, dtype=np.int64)  # (instance index, length)
    for k, (i, p) in enumerate(local):
        host[k, : len(p)] = np.frombuffer(p, dtype=np.uint8)
        host[k, len(p):] = 0
        hmeta[k] = (i, len(p))
    host[len(local):] = 0
    if buf is not hbuf:
        buf.copy_(hbuf, non_blocking=True)  # pinned -> device, ordered before the collective on the current stream
    meta = torch.from_numpy(hmeta).to(device)
    bufs = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    metas = [torch.empty_like(meta) for _ in range(world)] if rank == 0 else None
    works = [dist.gather(buf, bufs, dst=0, async_op=True), dist.gather(meta, metas, dst=0, async_op=True)]
    return _PendingGather(works, bufs, metas, n_total, (buf, meta))


def gather_proofs(local, n_total, device="cpu"):
    """Synchronous form of gather_proofs_async."""
    return gather_proofs_async(local, n_total, device).wait()
