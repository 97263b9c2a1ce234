"""Device memory comes back: provers, pools, trees and transforms created and destroyed many times leave the card's free memory where the
first cycles left it.  (What a context keeps on purpose — twiddle tables, jump matrices, per-stream workspaces — is allocated by the
warm-up cycles and is not a leak; an arena, a tree's digest layers, a pinned staging buffer or a stream that is not released shows as a
steady loss per cycle.)  The reference creates and drops its GPU objects per call (native/src/fib_air.rs:56-72: a prover per `run_fib_air_zk`)."""
import gc

import pytest

pytestmark = pytest.mark.gpu
MIB = 1 << 20


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _cycle(p3, k):
    import torch
    fp = p3.FriParameters(1, 0, 6, 4)
    pr = p3.FibAirProver(13, params=fp)
    pr.prove(k, k + 1)
    pr.close()
    pr = p3.FibAirProver(11, params=fp, hash="keccak", hiding=True, seed=1, profile="throughput")
    pr.prove(k, k + 1)
    pr.close()
    pr = p3.FibAirProver(3, params=p3.FriParameters(2, 2, 2, 1), hash="keccak", hiding=True, seed=1)  # the one-launch prover
    pr.prove(0, 1)
    pr.close()
    pool = p3.FibAirBatchProver(10, n_provers=2, params=fp)
    pool.prove([(k, k + 1), (k + 1, k + 2), (k + 2, k + 3)])
    pool.close()
    x = p3.dev_u32(p3.benchmark_input(1 << 12, 6))
    dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
    lde = dft.coset_lde_batch(x, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    for h in ("poseidon2", "keccak"):
        mmcs = p3.MerkleTreeMmcs(h)
        root, tree = mmcs.commit([lde])
        mmcs.open_batch(5, tree)
        tree.free()
    del x, lde, dft
    gc.collect()
    torch.cuda.empty_cache()


def test_device_memory_returns_after_create_destroy_cycles(p3):
    ok, msg = p3.is_available()
    assert ok, msg
    for k in range(3):  # warm-up: tables, workspaces and the caching allocator's pools reach their steady size
        _cycle(p3, k)
    import psutil
    me = psutil.Process()
    base, rss0 = _free_bytes(), me.memory_info().rss
    for k in range(25):
        _cycle(p3, 10 + k)
    lost, grown = base - _free_bytes(), me.memory_info().rss - rss0
    # 25 cycles x (3 provers + a pool of 2 + 2 trees): a leaked arena (>= 1 MiB at these sizes) or layer buffer would cost >= 25 MiB
    assert lost < 8 * MIB, "free device memory fell by %.1f MiB over 25 create / destroy cycles" % (lost / MIB)
    # host side: the pinned staging buffers (two per prover, ~0.1-1 MiB each here) and the proof vectors
    assert grown < 64 * MIB, "resident host memory grew by %.1f MiB over 25 create / destroy cycles" % (grown / MIB)
    print("device memory lost %.2f MiB, host RSS grown %.2f MiB over 25 cycles" % (lost / MIB, grown / MIB))
