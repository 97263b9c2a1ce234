import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from __graft_entry__ import load_package
p3 = load_package()
for n_provers in (4, 8):
    pool = p3.FibAirBatchProver(20, n_provers=n_provers)
    pool.prove([(i, i + 1) for i in range(16)])
    t0 = time.perf_counter()
    n = 0
    for k in range(4):
        pool.prove([(k * 64 + i, k * 64 + i + 1) for i in range(64)]); n += 64
    dt = time.perf_counter() - t0
    print("C batch pool, %d provers: %.1f proofs/s (4 batches of 64)" % (n_provers, n / dt))
    t0 = time.perf_counter()
    tickets, n = [pool.submit([(i, i + 1) for i in range(64)])], 64
    for k in range(1, 6):
        tickets.append(pool.submit([(k * 64 + i, k * 64 + i + 1) for i in range(64)])); n += 64
        pool.collect(tickets.pop(0))
    pool.collect(tickets.pop(0))
    dt = time.perf_counter() - t0
    print("C batch pool, %d provers, submit/collect one batch ahead: %.1f proofs/s (6 batches of 64)" % (n_provers, n / dt))
    pool.close()
