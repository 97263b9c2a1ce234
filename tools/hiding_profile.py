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
pr = p3.FibAirProver(log_n, params=p3.FriParameters(), hash=hash, hiding=True)
for i in range(6):
    pr.prove(i, i + 1)
pr.close()
print("done")
