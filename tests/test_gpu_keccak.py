"""GPU parity tests of the Keccak hash configuration (the one the reference itself wires, native/src/fib_air.rs:28-51)
through the C ABI, bit for bit against oracle/keccak.c, whose permutation is pinned by hashlib (test_oracle_keccak.py)."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0x78000001


def test_keccak_f_states(p3, oracle):
    rng = np.random.default_rng(25)
    st = rng.integers(0, 2**63, (1000, 25), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (1000, 25), dtype=np.uint64)
    st[0] = 0
    st[1] = np.uint64(0xFFFFFFFFFFFFFFFF)
    got = p3.keccak_f(st)
    for i in list(range(8)) + [500, 999]:
        assert np.array_equal(got[i], oracle.keccak_f(st[i])), i
    # the zero state's image begins with the well-known lane 0xF1258F7940E1DDE7
    assert int(got[0][0]) == 0xF1258F7940E1DDE7


def test_sha3_256_through_the_gpu_permutation(p3):
    """FIPS 202 SHA3-256 of a short message with the absorb/pad done here and the permutation on the GPU."""
    msg = b"Plonky3-mobile fib_air on MI355X"
    blk = bytearray(msg) + bytes([0x06]) + bytes(136 - len(msg) - 1)
    blk[-1] |= 0x80
    st = np.zeros((1, 25), dtype=np.uint64)
    st[0, :17] = np.frombuffer(bytes(blk), dtype="<u8")
    out = p3.keccak_f(st)
    assert out[0].tobytes()[:32] == hashlib.sha3_256(msg).digest()


@pytest.mark.parametrize("dims", [[(1, 2)], [(8, 2)], [(1 << 10, 2)], [(1 << 12, 4), (1 << 12, 33)], [(1 << 11, 36)],
                                  [(1 << 13, 2), (1 << 10, 5), (1 << 10, 70), (4, 3), (1, 9)], [(1 << 16, 2)],
                                  [(1 << 9, 35), (1 << 9, 1)]])
def test_keccak_mmcs_commit_open(p3, oracle, dims):
    rng = np.random.default_rng(sum(h * w for h, w in dims))
    mats = [rng.integers(0, P, d, dtype=np.uint32) for d in dims]
    mm = p3.MerkleTreeMmcs(hash="keccak")
    root, tree = mm.commit(mats)
    oroot, otree = oracle.mmcs_commit(mats, oracle.HASH_KECCAK)
    assert np.array_equal(root, oroot)
    for gl, ol in zip(tree.digest_layers(), otree.layers()):
        assert np.array_equal(gl, ol)
    maxh = max(h for h, _ in dims)
    for index in sorted({0, maxh - 1, maxh // 3, (5 * maxh) // 7}):
        rows, path = mm.open_batch(index, tree)
        orows, opath = otree.open_batch(index)
        assert np.array_equal(np.concatenate(rows) if rows else np.zeros(0, np.uint32), orows)
        assert np.array_equal(path, opath)
        assert oracle.mmcs_verify_batch(root, dims, index, orows, path, oracle.HASH_KECCAK)
    tree.free()


def test_keccak_tree_of_the_fib_air_lde(p3, oracle):
    """The commitment the reference's config would make over the fib_air trace LDE (non-hiding): 2^16-row trace,
    blowup 2, committed in bit-reversed order; root and a few openings against the oracle."""
    n = 1 << 16
    trace = p3.generate_trace_rows(0, 1, n)
    lde = p3.GpuDft.with_backend(p3.BackendKind.Hip).coset_lde_batch(trace, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    mm = p3.MerkleTreeMmcs(hash="keccak")
    root, tree = mm.commit([lde])
    host = oracle.coset_lde_batch(oracle.generate_trace_rows(0, 1, n), 1, p3.GENERATOR_MONTY, True)
    oroot, _ = oracle.mmcs_commit([host], oracle.HASH_KECCAK)
    assert np.array_equal(root, oroot)
    rows, path = mm.open_batch(12345, tree)
    assert oracle.mmcs_verify_batch(root, [(2 * n, 2)], 12345, rows[0], path, oracle.HASH_KECCAK)
    with pytest.raises(ValueError):
        p3.MerkleTreeMmcs(hash="sha256")


@pytest.mark.parametrize("log_n,t", [(1, (1, 0, 8, 4)), (3, (1, 0, 10, 4)), (8, (1, 0, 30, 8)), (10, (2, 1, 12, 10)),
                                     (12, (1, 0, 100, 16)), (14, (1, 3, 20, 12)), (11, (1, 4, 9, 7))])
def test_keccak_config_proof_bytes_equal_the_oracle(p3, oracle, log_n, t):
    """fib_air proved on the GPU under the reference's own hashes: proof bytes identical to the oracle's, accepted by
    both verifiers.  (1, 4, ..) and (1, 3, ..) make the pending transcript bytes at grind time cross a 136-byte
    block boundary: 32 chaining + 16 * 2^lfp final-polynomial bytes.)"""
    K = oracle.HASH_KECCAK
    gfp, ofp = p3.FriParameters(*t), oracle.FriParams(*t)
    prover = p3.FibAirProver(log_n, params=gfp, hash="keccak")
    try:
        for a in (0, 7):
            proof = prover.prove(a, a + 1)
            assert proof == oracle.prove_fib_air(a, a + 1, log_n, ofp, hash=K)
            x = oracle.fib_public_x(a, a + 1, 1 << log_n)
            assert oracle.verify_fib_air(proof, a, a + 1, x, log_n, ofp, hash=K) == 0
            p3.verify_fib_air(proof, a, a + 1, x, log_n, gfp, hash="keccak")
    finally:
        prover.close()


def test_keccak_config_headline_size(p3, oracle):
    """2^20 rows, benchmark FRI parameters, the reference's own hashes: the COMPLETE proof bytes equal the oracle prover's
    (all host cores; until round 5 only bench.py's cpu_baseline leg compared them at this size), and the oracle's verifier accepts."""
    prover = p3.FibAirProver(20, hash="keccak")
    try:
        proof = prover.prove(0, 1)
        assert prover.prove(0, 1) == proof
    finally:
        prover.close()
    x = oracle.fib_public_x(0, 1, 1 << 20)
    assert oracle.verify_fib_air(proof, 0, 1, x, 20, oracle.FriParams(), hash=oracle.HASH_KECCAK) == 0
    oracle.set_threads(oracle.test_threads())
    try:
        ref = oracle.prove_fib_air(0, 1, 20, oracle.FriParams(), hash=oracle.HASH_KECCAK)
    finally:
        oracle.set_threads(1)
    assert len(proof) == len(ref)
    if proof != ref:
        w1, w2 = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
        pytest.fail("cfg2 keccak: proof words differ first at %d of %d" % (int(np.nonzero(w1 != w2)[0][0]), len(w1)))
    assert p3.run_fib_air(hash="keccak") == "fib_air ok (n=8, x=21)"


def test_keccak_config_cfg3_size_proof_bytes_equal_oracle(p3, oracle):
    """BASELINE configs[2]'s size (2^24-row trace, blowup 4, the bench's 100 queries / 16 bits) under the reference's own hashes: the
    complete proof bytes — Keccak trees of 2^26 leaves, the Keccak-256 hash challenger over 24 FRI rounds — equal the oracle prover's
    (16 threads), and the oracle verifier accepts them."""
    t = (2, 0, 100, 16)
    prover = p3.FibAirProver(24, params=p3.FriParameters(*t), hash="keccak")
    try:
        proof = prover.prove(0, 1)
    finally:
        prover.close()
    x = oracle.fib_public_x(0, 1, 1 << 24)
    assert oracle.verify_fib_air(proof, 0, 1, x, 24, oracle.FriParams(*t), hash=oracle.HASH_KECCAK) == 0
    oracle.set_threads(oracle.test_threads())
    try:
        ref = oracle.prove_fib_air(0, 1, 24, oracle.FriParams(*t), hash=oracle.HASH_KECCAK)
    finally:
        oracle.set_threads(1)
    assert len(proof) == len(ref)
    if proof != ref:
        w1, w2 = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
        pytest.fail("cfg3 keccak: proof words differ first at %d of %d" % (int(np.nonzero(w1 != w2)[0][0]), len(w1)))


def test_keccak_batch_pool(p3, oracle):
    pool = p3.FibAirBatchProver(10, n_provers=3, params=p3.FriParameters(1, 0, 10, 6), hash="keccak")
    try:
        inst = [(i, i + 1) for i in range(7)]
        proofs = pool.prove(inst)
        ofp = oracle.FriParams(1, 0, 10, 6)
        for (a, b), pf in zip(inst, proofs):
            assert pf == oracle.prove_fib_air(a, b, 10, ofp, hash=oracle.HASH_KECCAK)
    finally:
        pool.close()
