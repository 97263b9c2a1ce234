"""CPU tests of the oracle's HIDING prover/verifier pair (oracle/stark_hiding.c: MerkleTreeHidingMmcs + HidingFriPcs as
the reference configures them, native/src/fib_air.rs:40-65) and of the random stream behind it (oracle/rng.c).
Upstream parity is UNPINNED (the crates are absent, the reference holds no fixture); pinned here: the generator
itself against its published reference vector, and the self-consistency of the protocol — an independently written
verifier accepts, rejects perturbed proofs and statements, and the blinding really changes what the proof reveals."""
import numpy as np
import pytest

P = 0x78000001


def test_xoshiro256pp_reference_vector(oracle):
    # rand_xoshiro's own test of Xoshiro256PlusPlus (state 1, 2, 3, 4), values from the generator's reference C code
    import ctypes as C
    s = (C.c_uint64 * 4)(1, 2, 3, 4)
    got = [oracle.rng_next_u64(s) for _ in range(10)]
    assert got == [41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205, 9973669472204895162,
                   14011001112246962877, 12406186145184390807, 15849039046786891736, 10450023813501588000]


def test_seed_from_u64_is_splitmix64(oracle):
    s = oracle.rng_seed_from_u64(1)
    assert s[0] == 0x910A2DEC89025CC1  # SplitMix64's first output for state 1 (published)
    # python restatement of SplitMix64 for all four words and another seed
    M = (1 << 64) - 1

    def splitmix(state, n):
        out = []
        for _ in range(n):
            state = (state + 0x9E3779B97F4A7C15) & M
            z = state
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            out.append(z ^ (z >> 31))
        return out
    assert list(s) == splitmix(1, 4)
    assert list(oracle.rng_seed_from_u64(0xDEADBEEF)) == splitmix(0xDEADBEEF, 4)


def test_field_sampling_is_rejection_of_31_bit_words(oracle):
    s1, s2 = oracle.rng_seed_from_u64(1), oracle.rng_seed_from_u64(1)
    got = oracle.rng_fill_field(s1, 2000)
    exp = []
    while len(exp) < 2000:
        v = (oracle.rng_next_u64(s2) >> 32) >> 1
        if v < P:
            exp.append(v)
    assert got.tolist() == exp and list(s1) == list(s2)  # same stream position afterwards
    assert int(got.max()) < P


@pytest.mark.parametrize("hash", [0, 1])
@pytest.mark.parametrize("log_n,t", [(1, (1, 0, 4, 2)), (3, (2, 2, 2, 1)), (6, (1, 0, 9, 5)), (9, (2, 1, 7, 6))])
def test_hiding_prove_then_verify(oracle, hash, log_n, t):
    fp = oracle.FriParams(*t)
    proof = oracle.prove_fib_air_hiding(3, 4, log_n, fp, hash=hash)
    x = oracle.fib_public_x(3, 4, 1 << log_n)
    assert oracle.verify_fib_air_hiding(proof, 3, 4, x, log_n, fp, hash=hash) == 0
    assert oracle.prove_fib_air_hiding(3, 4, log_n, fp, hash=hash) == proof          # deterministic for a seed
    assert oracle.verify_fib_air_hiding(proof, 3, 4, x + 1, log_n, fp, hash=hash) == 10  # OodEvaluationMismatch
    assert oracle.verify_fib_air_hiding(proof, 4, 4, x, log_n, fp, hash=hash) != 0
    assert oracle.verify_fib_air_hiding(proof, 3, 4, x, log_n, fp, hash=1 - hash) != 0  # other hash configuration


def test_reference_instance_and_parameters(oracle):
    # the reference's call: n = 8, x = 21, create_test_fri_params(mmcs, 2), both MMCS and PCS seeded with 1, Keccak hashes
    fp = oracle.FriParams(2, 2, 2, 1)
    proof = oracle.prove_fib_air_hiding(0, 1, 3, fp, hash=oracle.HASH_KECCAK, seed=1)
    assert oracle.verify_fib_air_hiding(proof, 0, 1, 21, 3, fp, hash=oracle.HASH_KECCAK) == 0
    assert oracle.verify_fib_air_hiding(proof, 0, 1, 22, 3, fp, hash=oracle.HASH_KECCAK) == 10


def test_every_tampered_word_is_rejected(oracle):
    fp = oracle.FriParams(1, 0, 3, 4)
    proof = oracle.prove_fib_air_hiding(0, 1, 4, fp)
    x = oracle.fib_public_x(0, 1, 16)
    words = np.frombuffer(proof, dtype=np.uint32)
    rng = np.random.default_rng(0)
    for pos in rng.choice(len(words), size=120, replace=False):
        bad = words.copy()
        bad[pos] = (int(bad[pos]) + 1) % P
        assert oracle.verify_fib_air_hiding(bad.tobytes(), 0, 1, x, 4, fp) != 0, pos
    for cut in (1, 4, 40, len(proof) // 2):
        assert oracle.verify_fib_air_hiding(proof[:-cut], 0, 1, x, 4, fp) != 0


def test_blinding_changes_everything_the_proof_reveals(oracle):
    """Two seeds, same statement: both verify, and no commitment or opened value coincides (the non-hiding proof of the
    same statement is a fixed function of it)."""
    fp = oracle.FriParams(1, 0, 5, 3)
    p1 = oracle.prove_fib_air_hiding(0, 1, 5, fp, seed=1)
    p2 = oracle.prove_fib_air_hiding(0, 1, 5, fp, seed=2)
    x = oracle.fib_public_x(0, 1, 32)
    assert oracle.verify_fib_air_hiding(p1, 0, 1, x, 5, fp) == 0 and oracle.verify_fib_air_hiding(p2, 0, 1, x, 5, fp) == 0
    w1, w2 = np.frombuffer(p1, np.uint32), np.frombuffer(p2, np.uint32)
    head = 3 + 24 + (1 + 32) + (1 + 24) * 2 + 1 + 4 * 17  # header, three roots, opened values of the three rounds
    assert len(w1) == len(w2)
    same = np.nonzero(w1[3:head] == w2[3:head])[0]
    # only the length prefixes coincide
    assert len(same) <= 8, same
