import sys, time
sys.path.insert(0, ".")
from __graft_entry__ import load_package
p3 = load_package()
import torch
for nprov in (4, 8, 12):
    pool = p3.FibAirBatchProver(20, n_provers=nprov)
    inst = [(i, i + 1) for i in range(64)]
    pool.prove(inst[:16])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        pool.prove(inst)
    dt = time.perf_counter() - t
    print("C pool provers=%d: %.1f proofs/s" % (nprov, 3 * 64 / dt))
    pool.close()
