#!/usr/bin/env python3
"""Leak soak on the GPU box: prover pools created and destroyed over and over (the pool's worker threads release their
per-thread contexts — tables, scratch — on exit), single provers of every flavour likewise, device memory watched with
hipMemGetInfo between cycles.   python tools/soak_pools.py [cycles] [log_n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
fp = p3.FriParameters(1, 0, 20, 8)


def free_mib():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] >> 20


base = None  # taken after the first full cycle: what the main thread's context and the HIP runtime keep for good
worst = 0
for c in range(cycles):
    for hash_cfg in ("poseidon2", "keccak"):
        pool = p3.FibAirBatchProver(log_n, n_provers=6, params=fp, hash=hash_cfg)
        proofs = pool.prove([(i, i + 1) for i in range(12)])
        p3.verify_fib_air(proofs[5], 5, 6, p3.fib_public_x(5, 6, 1 << log_n), log_n, fp, hash=hash_cfg)
        pool.close()
        for hiding in (False, True):
            pr = p3.FibAirProver(log_n - 2, params=fp, hash=hash_cfg, hiding=hiding)
            pf = pr.prove(c, c + 1)
            p3.verify_fib_air(pf, c, c + 1, p3.fib_public_x(c, c + 1, 1 << (log_n - 2)), log_n - 2, fp, hash=hash_cfg, hiding=hiding)
            pr.close()
    if base is None:
        base = free_mib()
        print("cycle  0: baseline taken, %d MiB free" % base, flush=True)
        continue
    delta = base - free_mib()
    worst = max(worst, delta)
    print("cycle %2d: device memory delta vs the first cycle %+d MiB" % (c, delta), flush=True)
# a leak of one worker's context per pool would be ~6 x (tables + scratch) per cycle: tens of MiB each
assert worst < 64, "device memory grows across pool create/destroy cycles: %d MiB" % worst
print("pool soak ok (max delta %d MiB over %d cycles)" % (worst, cycles))
