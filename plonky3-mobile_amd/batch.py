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

    def __init__(self, works, bufs, metas, n_total, keep, landed=None, event=None):
        self.works, self.bufs, self.metas, self.n_total, self.keep = works, bufs, metas, n_total, keep
        self.landed, self.event = landed, event

    def wait(self, copy=True):
        """Rank 0: all n_total proofs in instance order — bytes objects, or with copy=False uint8 views of the
        receive buffer (valid until the gather after next reuses it).  Other ranks: None."""
        if self.event is not None:
            self.event.synchronize()  # the collectives and the device -> pinned host copies behind them
        else:
            for w in self.works:
                w.wait()
        if self.bufs is None:
            return None
        out = [None] * self.n_total
        bufs, metas = (self.landed if self.landed is not None else (self.bufs, self.metas))
        for b, m in zip(bufs, metas):
            b, m = b.numpy(), m.tolist()
            for k, (i, ln) in enumerate(m):
                if i >= 0:
                    out[i] = b[k, :ln].tobytes() if copy else b[k, :ln]
        return out


class ProofGatherer:
    """Staging for the per-step gather of proof bytes to rank 0: three slots used round-robin, so that step k + 1 can be
    filling one while step k's gather is in flight and step k - 1's proofs are still being read on rank 0.

    open(n_local, width) -> (sink, slot): the prover threads call sink(row, index, proof) the moment a proof is
    serialised (the copy into the pinned buffer overlaps the other provers' GPU work); launch(slot) enqueues ONE H2D
    copy and the two gathers and returns at once.  On a GPU backend rank 0 also enqueues the copies of the received buffers
    into pinned host memory and an event behind them, so wait() costs no device round trip on the critical path.
    `width` (bytes per proof slot) must be the same on every rank: proofs of one parameter set have one length, so
    the caller learns it from the first step (gather_proofs_async does the all_reduce) and passes it from then on."""

    def __init__(self, n_total, device="cpu"):
        self.n_total, self.device = n_total, str(device)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.per = (n_total + self.world - 1) // self.world
        self.slots, self.turn = {}, 0

    def _slot(self, width):
        self.turn = (self.turn + 1) % 3
        key = (width, self.turn)
        ent = self.slots.get(key)
        if ent is None:
            gpu = self.device != "cpu"
            h = torch.zeros((self.per, width), dtype=torch.uint8)
            hm = torch.full((self.per, 2), -1, dtype=torch.int64)
            ent = {"h": h.pin_memory() if gpu else h, "hm": hm.pin_memory() if gpu else hm}
            if gpu:
                ent["d"] = torch.empty((self.per, width), dtype=torch.uint8, device=self.device)
                ent["dm"] = torch.empty((self.per, 2), dtype=torch.int64, device=self.device)
                if self.rank == 0:
                    ent["rb"] = [torch.empty_like(ent["d"]) for _ in range(self.world)]
                    ent["rm"] = [torch.empty_like(ent["dm"]) for _ in range(self.world)]
                    ent["lb"] = [torch.empty((self.per, width), dtype=torch.uint8).pin_memory() for _ in range(self.world)]
                    ent["lm"] = [torch.empty((self.per, 2), dtype=torch.int64).pin_memory() for _ in range(self.world)]
            elif self.rank == 0:
                ent["rb"] = [torch.empty_like(ent["h"]) for _ in range(self.world)]
                ent["rm"] = [torch.empty_like(ent["hm"]) for _ in range(self.world)]
            self.slots[key] = ent
        return ent

    def open(self, n_local, width):
        if n_local > self.per:
            raise ValueError("%d local proofs for %d slots" % (n_local, self.per))
        ent = self._slot(width)
        host, meta = ent["h"].numpy(), ent["hm"].numpy()
        meta[:] = -1

        def sink(row, index, proof):
            if len(proof) > width:
                raise ValueError("proof of %d bytes does not fit the agreed slot of %d" % (len(proof), width))
            host[row, : len(proof)] = np.frombuffer(proof, dtype=np.uint8)
            meta[row] = (index, len(proof))
        # direct form: the prover writes the proof into the staging row itself (FibAirProver.prove_into) — no bytes object, no copy
        base, stride = ent["h"].data_ptr(), ent["h"].stride(0)
        sink.row_ptr = lambda row: (base + row * stride, width)
        sink.set = lambda row, index, length: meta.__setitem__(row, (index, length))
        return sink, ent

    def launch(self, ent):
        if self.device == "cpu":
            buf, meta = ent["h"], ent["hm"]
        else:
            buf, meta = ent["d"], ent["dm"]
            buf.copy_(ent["h"], non_blocking=True)
            meta.copy_(ent["hm"], non_blocking=True)
        bufs, metas = (ent["rb"], ent["rm"]) if self.rank == 0 else (None, None)
        works = [dist.gather(buf, bufs, dst=0, async_op=True), dist.gather(meta, metas, dst=0, async_op=True)]
        landed = event = None
        if self.device != "cpu":
            for w in works:
                w.wait()  # stream-ordered: the current stream waits, the host does not
            if self.rank == 0:
                for src, dst in zip(bufs + metas, ent["lb"] + ent["lm"]):
                    dst.copy_(src, non_blocking=True)
                landed = (ent["lb"], ent["lm"])
            event = torch.cuda.Event()
            event.record()
        return _PendingGather(works, bufs, metas, self.n_total, ent, landed, event)


_GATHERERS = {}


def gather_proofs_async(local, n_total, device="cpu", width=None):
    """local: list of (index, proof bytes) of this rank.  Starts the gather to rank 0 and returns a handle whose
    wait() gives, on rank 0, all n_total proofs in instance order (None elsewhere).  Without `width` the ranks
    first agree on the longest proof (one all_reduce); the collective has a fixed shape and, being asynchronous,
    overlaps the next step's proving."""
    if width is None:
        max_len = torch.tensor([max([len(p) for _, p in local], default=0)], dtype=torch.int64, device=device)
        dist.all_reduce(max_len, op=dist.ReduceOp.MAX)
        width = int(max_len.item())
    g = _GATHERERS.get((n_total, str(device)))
    if g is None:
        g = _GATHERERS[(n_total, str(device))] = ProofGatherer(n_total, device)
    sink, slot = g.open(len(local), width)
    for k, (i, p) in enumerate(local):
        sink(k, i, p)
    pend = g.launch(slot)
    pend.width = width
    return pend


def gather_proofs(local, n_total, device="cpu"):
    """Synchronous form of gather_proofs_async."""
    return gather_proofs_async(local, n_total, device).wait()
