#!/usr/bin/env python3
"""Wall time of ONE proof on ONE prover, nothing else on the chip (mean of the later of `reps` back-to-back proofs):
   python3 tools/single_proof_latency.py [log_n=20] [hash=poseidon2] [hiding=0] [reps=12] [profile=latency]
No environment variable selects anything: the profile is an argument of the prover's creation (include/p3hip.h PROFILES)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
hash = sys.argv[2] if len(sys.argv) > 2 else "poseidon2"
hiding = len(sys.argv) > 3 and sys.argv[3] == "1"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
profile = sys.argv[5] if len(sys.argv) > 5 else "latency"
params = p3.FriParameters(2, 2, 2, 1) if log_n == 3 else p3.FriParameters()
pr = p3.FibAirProver(log_n, params=params, hash=hash, hiding=hiding, profile=profile)
for i in range(3):
    pr.prove(i, i + 1)
ts = []
for i in range(reps):
    t0 = time.perf_counter()
    pr.prove(10 + i, 11 + i)
    ts.append((time.perf_counter() - t0) * 1e3)
pr.close()
ts.sort()
print("single proof, 2^%d rows, %s%s, profile %s: median %.3f ms, min %.3f ms over %d proofs" % (
    log_n, hash, " + hiding" if hiding else "", profile, ts[len(ts) // 2], ts[0], reps))
