#!/usr/bin/env python3
"""Workload for a rocprofv3 PMC pass over the Keccak kernels (profiles/r02_pmc_keccak.json): Merkle commits under the
reference's own hashes at the prover's sizes — 2^21 x 2 (the trace tree of the 2^20 proof), 2^20 x 8 (FRI round 0) and
2^24 x 2.  Counters and formulas: tools/pmc_poseidon2_summarize.py."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
P = 0x78000001
mm = p3.MerkleTreeMmcs(hash="keccak")
for rep in range(2):  # the summariser takes the LAST launch of each kernel shape
    for log_h, w in ((21, 2), (20, 8), (24, 2)):
        x = torch.randint(0, P, (1 << log_h, w), dtype=torch.int32, device="cuda")
        root, tree = mm.commit([x])
        torch.cuda.synchronize()
        tree.free()
        del x
print("done")
