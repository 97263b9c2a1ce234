"""world_size-2 gloo test (CPU) of the multi-GPU batch plumbing: descriptor scatter, instance sharding and
proof gather.  The proving itself needs a GPU; here each rank 'proves' with a deterministic stand-in so the
collective layout is what is under test."""
import os
import socket
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_proof(i, a, b):
    return (b"proof-%d-%d-%d|" % (i, a, b)) * (3 + i % 4)  # ragged lengths


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    load_package()
    from plonky3_mobile_amd import batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    instances = [(i, i + 1) for i in range(7)] if rank == 0 else []
    mine = batch.scatter_descriptors(instances)
    assert [i for i, _, _ in mine] == batch.shard_instances(7, rank, world)
    local = [(i, _fake_proof(i, a, b)) for i, a, b in mine]
    allp = batch.gather_proofs(local, 7)
    if rank == 0:
        q.put(allp)
    dist.barrier()
    dist.destroy_process_group()


def test_scatter_prove_gather_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    allp = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert allp == [_fake_proof(i, i, i + 1) for i in range(7)]


def _fake_proof_step(step, i, a, b):
    # ragged, and the LONGEST proof changes rank and length from step to step (the staging width follows it)
    return (b"s%d-proof-%d-%d-%d|" % (step, i, a, b)) * (2 + (i * 7 + step * 5) % 9)


def _worker_async(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    load_package()
    from plonky3_mobile_amd import batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py's loop: the gather of step k is still in flight while step k+1 fills the OTHER staging pair and
    # starts its own gather; step k is collected only then.  Five steps reuse each of the two pairs at least twice.
    n_total, results, pending = 7, [], None
    for step in range(5):
        inst = [(10 * step + i, 10 * step + i + 1) for i in range(n_total)] if rank == 0 else []
        mine = batch.scatter_descriptors(inst)
        local = [(i, _fake_proof_step(step, i, a, b)) for i, a, b in mine]
        prev, pending = pending, batch.gather_proofs_async(local, n_total)
        if prev is not None:
            results.append(prev.wait())
    results.append(pending.wait())
    if rank == 0:
        q.put(results)
    dist.barrier()
    dist.destroy_process_group()


def test_async_gather_double_buffer_reuse_world2():
    """gather_proofs_async across five consecutive steps with ragged, step-dependent proof lengths: each step's
    proofs come back complete and in instance order although the next step's staging was filled meanwhile."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_async, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    results = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert len(results) == 5
    for step, allp in enumerate(results):
        assert allp == [_fake_proof_step(step, i, 10 * step + i, 10 * step + i + 1) for i in range(7)], step


def _worker_pipelined(rank, world, port, q):
    import threading
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    load_package()
    from plonky3_mobile_amd import batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # bench.py's steady state: step k + 1 is issued (scatter, staging slot opened, "prover" threads writing through the
    # sink) BEFORE step k retires (gather launched, gather k - 1 collected): collectives in the order scatter(k+1),
    # gather(k) on every rank, three staging slots in rotation, fixed slot width, rank 0 reads views.
    n_total, width, results, pending = 7, 64, [], None
    g = batch.ProofGatherer(n_total, "cpu")

    def descriptors(step):
        inst = [(10 * step + i, 10 * step + i + 1) for i in range(n_total)] if rank == 0 else []
        return batch.scatter_descriptors(inst)

    def issue(step):
        mine = descriptors(step)
        put, slot = g.open(len(mine), width)
        def write(r, i, a, b):
            data = b"S%d:%d:%d:%d" % (step, i, a, b) * (1 + i % 3)
            if step % 2 == 0:
                put(r, i, data)
            else:  # the direct form bench.py uses: the prover writes into the staging row itself (FibAirProver.prove_into)
                import ctypes
                address, cap = put.row_ptr(r)
                assert cap == width and len(data) <= cap
                ctypes.memmove(address, data, len(data))
                put.set(r, i, len(data))
        ths = [threading.Thread(target=write, args=(r, i, a, b)) for r, (i, a, b) in enumerate(mine)]
        [t.start() for t in ths]
        return ths, slot

    def retire(ths, slot):
        nonlocal pending
        [t.join() for t in ths]
        prev, pending = pending, g.launch(slot)
        if prev is not None:
            res = prev.wait(copy=False)
            results.append([bytes(x) for x in res] if res is not None else None)

    inflight = [issue(0)]
    for step in range(1, 7):
        inflight.append(issue(step))
        retire(*inflight.pop(0))
    retire(*inflight.pop(0))
    res = pending.wait(copy=False)
    results.append([bytes(x) for x in res] if res is not None else None)
    try:
        g.open(1, width)[0](0, 0, b"x" * (width + 1))
        overflow = "accepted"
    except ValueError:
        overflow = "refused"
    if rank == 0:
        q.put((results, overflow))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gatherer_world2():
    """ProofGatherer as bench.py drives it: prefetched scatter, sink called from threads, fixed slot width, zero-copy
    views on rank 0, seven overlapping steps over the three staging slots; an oversized proof is refused, not truncated."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_pipelined, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    results, overflow = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert overflow == "refused"
    assert len(results) == 7
    for step, allp in enumerate(results):
        assert allp == [b"S%d:%d:%d:%d" % (step, i, 10 * step + i, 10 * step + i + 1) * (1 + i % 3) for i in range(7)], step


def _worker_run_scatter(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    load_package()
    from plonky3_mobile_amd import batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a whole run's descriptors in ONE scatter (bench.py's form: every rank knows the shape), then the same with the shape
    # learnt from rank 0 and ragged steps (the last one shorter, one empty)
    steps = [[(100 * k + i, 100 * k + i + 1) for i in range(7)] for k in range(5)] if rank == 0 else []
    pend = batch.scatter_descriptor_steps(steps, shape=(5, 7))
    got = [pend.step(k) for k in (3, 0, 4, 1, 2)]  # any order, any number of times
    ragged = [[(1, 2), (3, 4), (5, 6)], [], [(9, 9)]] if rank == 0 else []
    pend2 = batch.scatter_descriptor_steps(ragged)
    got2 = [pend2.step(k) for k in range(len(pend2))]
    bad = None
    if rank == 0:
        try:
            batch.scatter_descriptor_steps([[(0, 1)] * 4], shape=(1, 3))
        except ValueError as e:
            bad = str(e)
    q.put((rank, got, got2, bad))
    dist.barrier()
    dist.destroy_process_group()


def test_one_scatter_for_a_whole_run_world2():
    """scatter_descriptor_steps: five steps' descriptors in one collective; rank r's shard of step k is instance i -> rank
    i mod world of THAT step; with and without the shape known to every rank; a shape that does not hold the steps is refused
    before anything is sent."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_run_scatter, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(2))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    for rank, got, got2, bad in res:
        for k, shard in zip((3, 0, 4, 1, 2), got):
            assert shard == [(i, 100 * k + i, 100 * k + i + 1) for i in range(rank, 7, 2)], (rank, k)
        ragged = [[(1, 2), (3, 4), (5, 6)], [], [(9, 9)]]
        assert got2 == [[(i, a, b) for i, (a, b) in enumerate(st) if i % 2 == rank] for st in ragged], rank
        assert (bad is not None and "announced" in bad) if rank == 0 else bad is None


def test_shard_instances_partition():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    load_package()
    from plonky3_mobile_amd import batch
    for world in (1, 2, 4, 8):
        got = sorted(i for r in range(world) for i in batch.shard_instances(64, r, world))
        assert got == list(range(64))
        assert all(len(batch.shard_instances(64, r, world)) == 64 // world for r in range(world))
