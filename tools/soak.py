#!/usr/bin/env python3
"""Soak test on the GPU box: many proofs through the batch pool, every one checked by the host verifier,
device memory watched for growth.  python tools/soak.py [n_proofs] [log_n] [poseidon2|keccak]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
n_proofs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hash_cfg = sys.argv[3] if len(sys.argv) > 3 else "poseidon2"
pool = p3.FibAirBatchProver(log_n, n_provers=8, hash=hash_cfg)
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
done = 0
xs = {}
while done < n_proofs:
    inst = [(done + i, done + i + 1) for i in range(64)]
    proofs = pool.prove(inst)
    for (a, b), pf in zip(inst[::8], proofs[::8]):  # verify a sample of each batch on the host
        x = p3.fib_public_x(a, b, 1 << log_n)
        p3.verify_fib_air(pf, a, b, x, log_n, hash=hash_cfg)
    done += len(inst)
    if done % 256 == 0:
        free = torch.cuda.mem_get_info()[0]
        print("%5d proofs, %.1f proofs/s incl. sampled verification, device memory delta %+d MiB" % (
            done, done / (time.perf_counter() - t0), (free0 - free) >> 20), flush=True)
pool.close()
print("soak ok")
