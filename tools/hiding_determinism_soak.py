#!/usr/bin/env python3
"""The hiding prover's side streams under load: N prover threads prove the SAME few instances over and over at two sizes; every proof
of an instance must have the same bytes as the first one (the streams restart from the seed for every proof), and the host verifier
must accept.  A missed cross-stream dependency (fills / randomization commitment racing their consumers) would show up as a differing
or rejected proof.   python3 tools/hiding_determinism_soak.py [threads=4] [rounds=6]   [profile=throughput|latency]
(the latency profile commits the randomization polynomial on a second side stream: round 4's P3HIP_HIDING_R_SIDE=1)"""
import os
import sys
import threading

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch  # noqa: E402,F401
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
profile = sys.argv[3] if len(sys.argv) > 3 else "throughput"
bad = []
for log_n, reps in ((14, 6 * rounds), (19, rounds)):
    fp = p3.FriParameters(1, 0, 20, 8)
    insts = [(0, 1), (3, 4), (9, 2)]
    ref = {}
    lock = threading.Lock()

    def worker(k):
        torch.cuda.set_device(0)
        pr = p3.FibAirProver(log_n, params=fp, hash="keccak", hiding=True, seed=1, profile=profile)
        for r in range(reps):
            a, b = insts[(r + k) % len(insts)]
            pf = pr.prove(a, b)
            with lock:
                first = ref.setdefault((a, b), pf)
            if pf != first:
                bad.append((log_n, k, r, a, b))
        pr.close()
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for (a, b), pf in ref.items():
        p3.verify_fib_air(pf, a, b, p3.fib_public_x(a, b, 1 << log_n), log_n, fp, hash="keccak", hiding=True)
    print("2^%d rows: %d provers x %d proofs, %d distinct instances, mismatches so far %d" % (log_n, threads, reps, len(ref), len(bad)), flush=True)
assert not bad, bad[:5]
print("hiding determinism soak ok")
