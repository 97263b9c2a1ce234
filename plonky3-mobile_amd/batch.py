"""Batches of independent fib_air proofs across the GPUs of one node (BASELINE configs[3]).

Proofs are independent, so there is no collective on the data path: instance i goes to rank i mod world.
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is used only to
scatter the instance descriptors (a, b) from rank 0 and to gather the proof bytes back.

Round 5: the collectives never share a queue with the provers.  On a GPU backend they run on a stream of their
own at the highest priority the device offers (`collective_stream`; `init_process_group_for_batches` also asks
RCCL for its high-priority internal stream), a whole run's descriptors travel in ONE scatter
(`scatter_descriptor_steps`: SURVEY.md §8e "one scatter in"), a step's proofs and their (index, length) headers
travel in ONE gather (the header is the first 16 bytes of every staging row), and the host never blocks on a
device round trip while issuing work: results are waited for through events recorded behind pinned copies."""
import numpy as np
import torch
import torch.distributed as dist

ROW_HEADER = 16  # bytes in front of every staging row: int64 instance index (-1 = empty row), int64 proof length


def shard_instances(n_total, rank, world):
    """Indices of the instances rank `rank` proves: i = rank, rank + world, ..."""
    return list(range(rank, n_total, world))


_STREAMS = {}


def collective_stream(device):
    """The stream the batch collectives and their staging copies run on: one per device, highest priority, so that
    RCCL's kernels and the copies are scheduled ahead of the provers' queued launches instead of behind them.
    None on the CPU (gloo) path."""
    if str(device) == "cpu":
        return None
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _STREAMS.get(idx)
    if s is None:
        try:
            hi = torch.cuda.Stream.priority_range()[1]  # (least, greatest): numerically lower = higher priority
        except Exception:  # noqa: BLE001 - older torch: -1 is "high"
            hi = -1
        s = _STREAMS[idx] = torch.cuda.Stream(device=idx, priority=min(hi, -1))
    return s


def init_process_group_for_batches(backend, local_rank=0):
    """init_process_group for the scatter / gather of batches.  "nccl" (= RCCL on ROCm): bound to this rank's device and with
    RCCL's own stream at high priority, so a collective issued while four provers saturate the compute queues does not wait
    for their launches to drain (round 4 measured 8.5 ms per 24-byte descriptor scatter at one rank).  Returns a short
    description for the bench line."""
    if backend != "nccl":
        dist.init_process_group(backend)
        return backend + " (rehearsal, not RCCL)"
    note = "rccl, process-group stream at high priority"
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), pg_options=opts)
    except (AttributeError, TypeError):
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        note = "rccl, default-priority process-group stream (this torch has no ProcessGroupNCCL.Options)"
    return note


class _Landing:
    """A small device -> pinned-host copy with an event behind it (no blocking .cpu() on the issue path)."""

    def __init__(self, dev_tensor, stream):
        self.host = torch.empty(dev_tensor.shape, dtype=dev_tensor.dtype).pin_memory()
        self.host.copy_(dev_tensor, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(stream)
        self.keep = dev_tensor

    def tolist(self):
        self.event.synchronize()
        return self.host.tolist()


def _on(stream):
    return torch.cuda.stream(stream) if stream is not None else _Null()


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


class PendingDescriptors:
    """The descriptors of several steps, scattered in one collective; step(k) gives this rank's shard of step k as a list of
    (index, a, b).  The first call waits for the one event behind the pinned copy."""

    def __init__(self, landing, steps, per):
        self._landing, self._steps, self._per, self._rows = landing, steps, per, None

    def step(self, k):
        if self._rows is None:
            self._rows = self._landing.tolist()
            self._landing = None
        rows = self._rows[k * self._per:(k + 1) * self._per]
        return [(int(i), int(a), int(b)) for i, a, b in rows if i >= 0]

    def __len__(self):
        return self._steps


class DescriptorScatter:
    """The scatter of a run's descriptors with every buffer allocated up front (pinned host table, device send / receive buffers,
    pinned landing buffer): run() then costs one table fill, one H2D copy, ONE scatter and one D2H copy, all enqueued on the
    collective stream — no allocation and no pinned-memory registration inside a timed region."""

    def __init__(self, max_steps, n, device="cpu"):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.device, self.stream = str(device), collective_stream(device)
        self.max_steps, self.n = max_steps, n
        self.per = (n + self.world - 1) // self.world
        rows = max(max_steps * self.per, 1)
        gpu = self.stream is not None
        self.recv_host = torch.full((rows, 3), -1, dtype=torch.int64)
        if gpu:
            self.recv_host = self.recv_host.pin_memory()
            self.recv_dev = torch.full((rows, 3), -1, dtype=torch.int64, device=device)
            self.event = torch.cuda.Event()
        if self.rank == 0:
            self.table = torch.full((self.world, rows, 3), -1, dtype=torch.int64)
            if gpu:
                self.table = self.table.pin_memory()
                self.table_dev = torch.empty((self.world, rows, 3), dtype=torch.int64, device=device)
        self.last_ms = {}

    def run(self, steps):
        """steps (rank 0; ignored elsewhere): one list of instances (a, b) per step, at most max_steps of at most n.  Returns
        PendingDescriptors; nothing blocks."""
        import time
        t0 = time.perf_counter()
        n_steps = len(steps) if self.rank == 0 else None
        if self.rank == 0:
            if n_steps > self.max_steps or any(len(s) > self.n for s in steps):
                raise ValueError("scatter_descriptor_steps: %d steps of at most %d instances announced, got %r" % (self.max_steps, self.n, [len(s) for s in steps]))
            tab = self.table.numpy()
            tab[:] = -1
            for k, inst in enumerate(steps):
                arr = np.asarray(inst, dtype=np.int64).reshape(len(inst), 2)
                for r in range(self.world):
                    idx = np.arange(r, len(inst), self.world, dtype=np.int64)
                    tab[r, k * self.per:k * self.per + len(idx), 0] = idx
                    tab[r, k * self.per:k * self.per + len(idx), 1:] = arr[idx]
        t1 = time.perf_counter()
        if self.stream is None:
            dist.scatter(self.recv_host, [self.table[r] for r in range(self.world)] if self.rank == 0 else None, src=0)
            self.last_ms = {"fill": 1e3 * (t1 - t0), "issue": 1e3 * (time.perf_counter() - t1)}
            return PendingDescriptors(self.recv_host, self.max_steps, self.per)
        with torch.cuda.stream(self.stream):
            if self.rank == 0:
                self.table_dev.copy_(self.table, non_blocking=True)
                dist.scatter(self.recv_dev, [self.table_dev[r] for r in range(self.world)], src=0)
            else:
                dist.scatter(self.recv_dev, None, src=0)
            self.recv_host.copy_(self.recv_dev, non_blocking=True)
            self.event.record(self.stream)
        self.last_ms = {"fill": 1e3 * (t1 - t0), "issue": 1e3 * (time.perf_counter() - t1)}
        return PendingDescriptors(_Landed(self.recv_host, self.event), self.max_steps, self.per)


class _Landed:
    def __init__(self, host, event):
        self.host, self.event = host, event

    def tolist(self):
        self.event.synchronize()
        return self.host.tolist()


def scatter_descriptor_steps(steps, device="cpu", shape=None):
    """`steps` (rank 0 only; anything on the other ranks): a list with one list of instances (a, b) per step.  ONE scatter moves
    every step's descriptors; rank r receives, for each step, the instances shard_instances(n, r, world) of that step.
    `shape` = (number of steps, instances per step) when every rank knows it (nothing but the scatter is issued then);
    without it rank 0's shape is broadcast first.  Returns PendingDescriptors.  (A caller that scatters inside a timed region
    keeps a DescriptorScatter instead: this convenience form allocates its buffers per call.)"""
    rank = dist.get_rank()
    stream = collective_stream(device)
    if shape is None:
        with _on(stream):
            t = torch.tensor([len(steps), max([len(s) for s in steps], default=0)] if rank == 0 else [0, 0], dtype=torch.int64)
            if stream is not None:
                t = t.pin_memory().to(device, non_blocking=True)
            dist.broadcast(t, 0)
            shape = (_Landing(t, stream).tolist() if stream is not None else t.tolist())
    n_steps, n = int(shape[0]), int(shape[1])
    if rank == 0 and (len(steps) != n_steps or any(len(s) > n for s in steps)):
        raise ValueError("scatter_descriptor_steps: %d steps of at most %d instances announced, got %r" % (n_steps, n, [len(s) for s in steps]))
    return DescriptorScatter(n_steps, n, device).run(steps)


def scatter_descriptors(instances, device="cpu"):
    """Rank 0 holds `instances` = list of (a, b); every rank receives its shard (list of (index, a, b)).  One step's worth;
    a run of several steps sends them all at once with scatter_descriptor_steps."""
    return scatter_descriptor_steps([instances] if dist.get_rank() == 0 else [], device).step(0)


class _PendingGather:
    """An in-flight gather of one step's proofs (one async collective); wait() returns the proofs on rank 0."""

    def __init__(self, work, rows, n_total, keep, event=None):
        self.work, self.rows, self.n_total, self.keep, self.event = work, rows, n_total, keep, event

    def wait(self, copy=True):
        """Rank 0: all n_total proofs in instance order — bytes objects, or with copy=False uint8 views of the
        receive buffer (valid until the gather after next reuses it).  Other ranks: None."""
        if self.event is not None:
            self.event.synchronize()  # the collective and the device -> pinned host copies behind it
        else:
            self.work.wait()
        if self.rows is None:
            return None
        out = [None] * self.n_total
        for b in self.rows:
            b = b.numpy()
            head = b[:, :ROW_HEADER].copy().view(np.int64)
            for k, (i, ln) in enumerate(head.tolist()):
                if i >= 0:
                    out[i] = b[k, ROW_HEADER:ROW_HEADER + ln].tobytes() if copy else b[k, ROW_HEADER:ROW_HEADER + ln]
        return out


class ProofGatherer:
    """Staging for the per-step gather of proof bytes to rank 0: three slots used round-robin, so that step k + 1 can be
    filling one while step k's gather is in flight and step k - 1's proofs are still being read on rank 0.

    open(n_local, width) -> (sink, slot): the prover threads call sink(row, index, proof) the moment a proof is
    serialised (the copy into the pinned buffer overlaps the other provers' GPU work); launch(slot) enqueues ONE H2D
    copy and ONE gather on the collective stream and returns at once.  On a GPU backend rank 0 also enqueues the copies of
    the received buffers into pinned host memory and an event behind them, so wait() costs no device round trip on the
    critical path.  A staging row is a 16-byte header (instance index, proof length) followed by `width` proof bytes, so
    a step is one collective.  `width` must be the same on every rank: proofs of one parameter set have one length, so
    the caller learns it from the first step (gather_proofs_async does the all_reduce) and passes it from then on."""

    def __init__(self, n_total, device="cpu", width=None):
        self.n_total, self.device = n_total, str(device)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.per = (n_total + self.world - 1) // self.world
        self.slots, self.turn = {}, 0
        self.stream = collective_stream(device)
        if width is not None:
            # all three staging slots now (pinned host rows, device rows, receive and landing buffers): a slot created on first use
            # page-locks tens of megabytes, which must not happen inside a timed region (round 5: the third slot was first used by the
            # first TIMED step after two warmup steps: ~90 ms of a 1.7 s run)
            for _ in range(3):
                self._slot(width)
            self.turn = 0

    def _slot(self, width):
        self.turn = (self.turn + 1) % 3
        key = (width, self.turn)
        ent = self.slots.get(key)
        if ent is None:
            gpu = self.device != "cpu"
            row = ROW_HEADER + ((width + 15) & ~15)
            h = torch.zeros((self.per, row), dtype=torch.uint8)
            ent = {"h": h.pin_memory() if gpu else h, "width": width}
            if gpu:
                ent["event"] = torch.cuda.Event()  # one per slot, re-recorded by every launch (creating / destroying an event per step
                                                   # costs the issuing thread ~1 ms a step under load: hipEventDestroy)
                ent["d"] = torch.empty((self.per, row), dtype=torch.uint8, device=self.device)
                if self.rank == 0:
                    ent["rb"] = [torch.empty_like(ent["d"]) for _ in range(self.world)]
                    ent["lb"] = [torch.empty((self.per, row), dtype=torch.uint8).pin_memory() for _ in range(self.world)]
            elif self.rank == 0:
                ent["rb"] = [torch.empty_like(ent["h"]) for _ in range(self.world)]
            self.slots[key] = ent
        return ent

    def open(self, n_local, width):
        if n_local > self.per:
            raise ValueError("%d local proofs for %d slots" % (n_local, self.per))
        ent = self._slot(width)
        host = ent["h"].numpy()
        head = host[:, :ROW_HEADER].view(np.int64)  # (per, 2): index, length
        head[:, 0] = -1
        head[:, 1] = 0

        def sink(row, index, proof):
            if len(proof) > width:
                raise ValueError("proof of %d bytes does not fit the agreed slot of %d" % (len(proof), width))
            host[row, ROW_HEADER:ROW_HEADER + len(proof)] = np.frombuffer(proof, dtype=np.uint8)
            head[row] = (index, len(proof))
        # direct form: the prover writes the proof into the staging row itself (FibAirProver.prove_into) — no bytes object, no copy
        base, stride = ent["h"].data_ptr(), ent["h"].stride(0)
        sink.row_ptr = lambda row: (base + row * stride + ROW_HEADER, width)
        sink.set = lambda row, index, length: head.__setitem__(row, (index, length))
        return sink, ent

    def launch(self, ent):
        if self.device == "cpu":
            rows = ent["rb"] if self.rank == 0 else None
            work = dist.gather(ent["h"], rows, dst=0, async_op=True)
            return _PendingGather(work, rows, self.n_total, ent)
        with torch.cuda.stream(self.stream):
            ent["d"].copy_(ent["h"], non_blocking=True)
            bufs = ent["rb"] if self.rank == 0 else None
            work = dist.gather(ent["d"], bufs, dst=0, async_op=True)
            work.wait()  # stream-ordered: the collective stream waits, the host does not
            rows = None
            if self.rank == 0:
                for src, dst in zip(bufs, ent["lb"]):
                    dst.copy_(src, non_blocking=True)
                rows = ent["lb"]
            event = ent["event"]
            event.record(self.stream)
        return _PendingGather(work, rows, self.n_total, ent, event)


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
