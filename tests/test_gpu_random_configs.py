"""A seeded sweep over the prover's whole parameter space at small sizes: trace height, blowup, final-polynomial length, query count,
proof-of-work bits, hash configuration, hiding or not, first trace row, generator seed, profile — drawn at random (fixed seed: the same 200
cases every run) — complete proof bytes against the oracle prover and the oracle verifier's verdict.  The fixed lists of the other files name
the cases somebody thought of; this one covers combinations nobody did.  (The reference has one instance: n = 8, native/src/fib_air.rs:56-72.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0x78000001


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        hiding = bool(rng.integers(0, 2))
        log_n = int(rng.integers(1, 15))
        log_blowup = int(rng.integers(1, 4))
        top = log_n + 1 if hiding else log_n  # log_final_poly_len stays below the (randomized) trace's log height
        log_fpl = 0 if rng.integers(0, 3) == 0 else int(rng.integers(0, top))
        queries = int(rng.integers(1, 25))
        pow_bits = int(rng.integers(0, 13))
        hash_name = ("poseidon2", "keccak")[int(rng.integers(0, 2))]
        a, b = int(rng.integers(0, P)), int(rng.integers(0, P))
        gen_seed = int(rng.integers(0, 1 << 40))
        profile = ("latency", "throughput")[i & 1]
        out.append((hiding, log_n, (log_blowup, log_fpl, queries, pow_bits), hash_name, a, b, gen_seed, profile))
    return out


def test_random_configurations_equal_the_oracle(p3, oracle):
    for case in _cases(200, 20261005):
        hiding, log_n, t, hash_name, a, b, gen_seed, profile = case
        kind = oracle.HASH_KECCAK if hash_name == "keccak" else oracle.HASH_POSEIDON2
        ofp = oracle.FriParams(*t)
        x = oracle.fib_public_x(a, b, 1 << log_n)
        if hiding:
            ref = oracle.prove_fib_air_hiding(a, b, log_n, ofp, hash=kind, seed=gen_seed)
            pr = p3.FibAirProver(log_n, params=p3.FriParameters(*t), hash=hash_name, hiding=True, seed=gen_seed, profile=profile)
        else:
            ref = oracle.prove_fib_air(a, b, log_n, ofp, hash=kind)
            pr = p3.FibAirProver(log_n, params=p3.FriParameters(*t), hash=hash_name, profile=profile)
        try:
            got = pr.prove(a, b)
        finally:
            pr.close()
        assert got == ref, case
        ok = (oracle.verify_fib_air_hiding(got, a, b, x, log_n, ofp, hash=kind) if hiding else oracle.verify_fib_air(got, a, b, x, log_n, ofp, hash=kind))
        assert ok == 0, case
