#!/usr/bin/env python3
"""N proofs on ONE prover, for rocprofv3 passes (kernel trace or PMC) over whole proofs:
   python3 tools/prove_n.py <hash> <log_n> <hiding 0|1> <n_proofs> [profile=throughput] [log_blowup=1]
FRI parameters are bench.py's (100 queries, 16 proof-of-work bits), the reference's (2, 2, 2, 1) for log_n = 3."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
hash = sys.argv[1] if len(sys.argv) > 1 else "poseidon2"
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
hiding = len(sys.argv) > 3 and sys.argv[3] == "1"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
profile = sys.argv[5] if len(sys.argv) > 5 else "throughput"
log_blowup = int(sys.argv[6]) if len(sys.argv) > 6 else 1
params = p3.FriParameters(2, 2, 2, 1) if log_n == 3 else p3.FriParameters(log_blowup=log_blowup)
pr = p3.FibAirProver(log_n, params=params, hash=hash, hiding=hiding, profile=profile)
for i in range(n):
    pr.prove(i, i + 1)
pr.close()
print("done: %d proofs" % n)
