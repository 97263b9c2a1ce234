"""Round 5 retired the experiment switches (41 environment variables, most of them selecting paths measured slower at every
size: DESIGN.md section 4.3 lists them).  What is left to select is (a) the PROFILE — throughput | latency, fixed when a prover
is created or set per thread for the free functions (include/p3hip.h PROFILES), so both run in ONE process here, byte for
byte against the oracle (round 4: nineteen child processes) — and (b) two test-only switches read when a prover is created
or proves (P3HIP_HIDING_PIECEWISE, P3HIP_GRIND_FIRST_LOG), set in-process by tests/test_gpu_hiding.py and tests/test_gpu_prover.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0x78000001


@pytest.mark.parametrize("profile", ["throughput", "latency"])
def test_trees_under_both_thread_profiles(p3, oracle, profile):
    """MMCS commits are free functions: they take the calling thread's profile.  Heights across every small-layer threshold
    (cooperative below 2^12 / 2^10, quad Poseidon2 and cooperative Keccak up to 2^15 / 2^12 under the latency profile), single
    and mixed-height matrix sets, and dense leaf rows of 9..64 words at 2^12..2^15 rows (the quad leaf kernel's multi-block
    sponge: the advisor's round-4 finding) — digest layers equal the oracle's, layer by layer, for both hashes."""
    rng = np.random.default_rng(11)
    assert p3.get_thread_profile() == "latency"  # the default of a thread that has not chosen
    p3.set_thread_profile(profile)
    try:
        assert p3.get_thread_profile() == profile
        for hash_name, kind in (("poseidon2", oracle.HASH_POSEIDON2), ("keccak", oracle.HASH_KECCAK)):
            shapes = [[(1 << 16, 2)], [(1 << 13, 3), (1 << 9, 5), (8, 2)], [(64, 9)], [(1 << 12, 9)], [(1 << 13, 17)], [(1 << 14, 33)],
                      [(1 << 15, 64)], [(1 << 12, 8)], [(1 << 15, 2)], [(1 << 10, 4)], [(1 << 11, 4)]]
            for dims in shapes:
                mats = [rng.integers(0, P, d, dtype=np.uint32) for d in dims]
                root, tree = p3.MerkleTreeMmcs(hash=hash_name).commit(mats)
                oroot, otree = oracle.mmcs_commit(mats, kind)
                assert np.array_equal(root, oroot), (profile, hash_name, dims)
                for gl, ol in zip(tree.digest_layers(), otree.layers()):
                    assert np.array_equal(gl, ol), (profile, hash_name, dims, len(gl))
                tree.free()
    finally:
        p3.set_thread_profile("latency")
    with pytest.raises(ValueError):
        p3.set_thread_profile("fastest")


def test_proofs_under_both_profiles_in_one_process(p3, oracle):
    """A throughput prover and a latency prover side by side in one process (round 4 needed one child process per setting): the
    same proof bytes as the oracle for both hashes, non-hiding and hiding; sizes whose FRI rounds are ALL tail (LDE of at most
    2^8 rows), proofs with a final polynomial, blowup 4 and 8, and sizes whose trees cross the quad / cooperative thresholds."""
    cases = [(1, (1, 0, 4, 2)), (3, (2, 2, 6, 5)), (6, (1, 0, 9, 5)), (7, (1, 3, 9, 0)), (9, (3, 1, 4, 10)), (10, (1, 0, 20, 8)),
             (12, (2, 0, 10, 4)), (13, (1, 0, 10, 6)), (15, (1, 0, 8, 5))]
    for hash_name, kind in (("poseidon2", oracle.HASH_POSEIDON2), ("keccak", oracle.HASH_KECCAK)):
        for log_n, t in cases:
            ref = oracle.prove_fib_air(3, 5, log_n, oracle.FriParams(*t), hash=kind)
            provers = [p3.FibAirProver(log_n, params=p3.FriParameters(*t), hash=hash_name, profile=pf) for pf in ("throughput", "latency")]
            for pr in provers:
                assert pr.prove(3, 5) == ref, (hash_name, log_n, t, pr.profile)
            for pr in provers:
                assert pr.prove(3, 5) == ref, (hash_name, log_n, t, pr.profile, "second proof")
                pr.close()
    for hash_name, kind in (("poseidon2", oracle.HASH_POSEIDON2), ("keccak", oracle.HASH_KECCAK)):
        for log_n, t in [(3, (2, 2, 2, 1)), (9, (1, 0, 8, 4)), (13, (1, 0, 6, 4))]:
            ref = oracle.prove_fib_air_hiding(0, 1, log_n, oracle.FriParams(*t), hash=kind, seed=1)
            for pf in ("throughput", "latency"):
                pr = p3.FibAirProver(log_n, params=p3.FriParameters(*t), hash=hash_name, hiding=True, seed=1, profile=pf)
                assert pr.prove(0, 1) == ref and pr.prove(0, 1) == ref, (hash_name, log_n, t, pf, "hiding")
                pr.close()
    with pytest.raises(ValueError):
        p3.FibAirProver(5, profile="fastest")


def test_hiding_proofs_just_above_the_one_launch_provers_sizes(p3, oracle):
    """Both hiding provers under both profiles at the sizes right above what the one-launch prover takes (an LDE of 2^9 .. 2^10 points, where almost
    every FRI round is a handful of rows), with and without a final polynomial, blowup 2 .. 8, a seed other than 1; and the Keccak non-hiding prover at
    such sizes.  (Round 5 also built these rounds as ONE launch for the Keccak / hiding provers — commit ce13ca9 — with equal bytes on exactly these
    cases; it measured neutral to slower, profiles/r05_fri_tail_any_ab.txt, and was taken out again.)"""
    cases = [(7, (1, 2, 5, 3)), (7, (2, 0, 4, 2)), (8, (1, 1, 3, 0)), (6, (3, 3, 3, 1)), (11, (1, 0, 6, 4))]
    for hash_name, kind in (("keccak", oracle.HASH_KECCAK), ("poseidon2", oracle.HASH_POSEIDON2)):
        for log_n, t in cases:
            ref = oracle.prove_fib_air_hiding(2, 7, log_n, oracle.FriParams(*t), hash=kind, seed=3)
            for pf in ("latency", "throughput"):
                pr = p3.FibAirProver(log_n, params=p3.FriParameters(*t), hash=hash_name, hiding=True, seed=3, profile=pf)
                assert pr.prove(2, 7) == ref and pr.prove(2, 7) == ref, (hash_name, log_n, t, pf)
                pr.close()
    for log_n, t in [(8, (1, 2, 5, 3)), (9, (2, 0, 4, 2)), (11, (1, 4, 3, 0)), (16, (1, 0, 6, 4))]:
        ref = oracle.prove_fib_air(2, 7, log_n, oracle.FriParams(*t), hash=oracle.HASH_KECCAK)
        pr = p3.FibAirProver(log_n, params=p3.FriParameters(*t), hash="keccak", profile="latency")
        assert pr.prove(2, 7) == ref and pr.prove(2, 7) == ref, (log_n, t)
        pr.close()


def test_pool_of_one_is_a_lone_prover(p3, oracle):
    """p3hip_fib_batch_create*: a pool of more than one prover runs the throughput profile, a pool of ONE the latency profile;
    the bytes are the oracle's either way."""
    t = (1, 0, 12, 5)
    ref = [oracle.prove_fib_air(a, a + 1, 13, oracle.FriParams(*t)) for a in range(3)]
    for n_provers in (1, 3):
        pool = p3.FibAirBatchProver(13, n_provers=n_provers, params=p3.FriParameters(*t))
        try:
            assert pool.prove([(a, a + 1) for a in range(3)]) == ref, n_provers
        finally:
            pool.close()
