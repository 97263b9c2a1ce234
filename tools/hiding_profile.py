#!/usr/bin/env python3
"""One hiding prover under rocprofv3 --kernel-trace --stats:  python3 tools/hiding_profile.py [hash] [log_n]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
hash = sys.argv[1] if len(sys.argv) > 1 else "poseidon2"
log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 19
# log_n = 3 is the reference's own instance: create_test_fri_params(challenge_mmcs, 2) (fib_air.rs:62)
params = p3.FriParameters(2, 2, 2, 1) if log_n == 3 else p3.FriParameters()
pr = p3.FibAirProver(log_n, params=params, hash=hash, hiding=True)
for i in range(6):
    pr.prove(i, i + 1)
pr.close()
print("done")
