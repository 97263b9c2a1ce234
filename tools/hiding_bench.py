#!/usr/bin/env python3
"""Timing of the HIDING prover (the reference's own configuration: Keccak hashes, MerkleTreeHidingMmcs, HidingFriPcs,
fib_air.rs:28-65) and of the Poseidon2 flavour, single prover, proofs accepted by the product's host verifier:
   python tools/hiding_bench.py [log_n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
sizes = [int(v) for v in sys.argv[1:]] or [3, 14, 18, 19]
for hash in ("keccak", "poseidon2"):
    for log_n in sizes:
        fp = p3.FriParameters(2, 2, 2, 1) if log_n == 3 else p3.FriParameters()
        pr = p3.FibAirProver(log_n, params=fp, hash=hash, hiding=True, seed=1)
        proof = pr.prove(0, 1)
        p3.verify_fib_air(proof, 0, 1, p3.fib_public_x(0, 1, 1 << log_n), log_n, fp, hash=hash, hiding=True)
        reps = 5
        t0 = time.perf_counter()
        for i in range(reps):
            pr.prove(i, i + 1)
        dt = (time.perf_counter() - t0) / reps
        print("hiding fib_air, %s hashes, 2^%d-row trace (randomized to 2^%d, LDE 2^%d): %.2f ms per proof, %d proof bytes, "
              "verified" % (hash, log_n, log_n + 1, log_n + 1 + fp.log_blowup, dt * 1e3, len(proof)))
        pr.close()
