"""GPU parity tests of the HIDING half of the reference's configuration (native/src/fib_air.rs:40-65): the device-resident
SmallRng streams, and the hiding prover whose proofs (randomized trace, blinded quotient chunks, randomization
polynomial, salted leaves) must equal the CPU oracle's byte for byte under both hash configurations.  Upstream parity of
the protocol is unpinned (oracle/stark_hiding.c); the generator itself is pinned (tests/test_oracle_hiding.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 0xDEADBEEF])
def test_device_rng_stream_equals_host_stream(p3, oracle, seed):
    """Fills of many sizes back to back: every element and the generator state afterwards equal a sequential host loop
    (the rejections make positions data-dependent; chunk boundaries at 1024 raw draws are crossed on purpose)."""
    import torch
    ok, msg = p3.is_available()
    assert ok, msg
    rng = p3.DeviceRng(seed)
    host = oracle.rng_seed_from_u64(seed)
    for n in [1, 3, 959, 960, 961, 1024, 5000, 0, 70001, 1 << 20, 17]:
        got = rng.fill_field(n)
        torch.cuda.synchronize()
        exp = oracle.rng_fill_field(host, n)
        assert np.array_equal(p3.host_u32(got), exp), n
        assert rng.state() == list(host), n
    rng.close()


def test_device_rng_far_jumps_equal_host_stream(p3, oracle):
    """Fills as long as the ones the 2^20-row hiding prover issues (pcs stream ~1.0e8 draws, mmcs ~4e7, fri ~1.6e7; here
    1.2e8, 5e7 and 1.7e7 back to back on ONE stream, so the second and third start from a state left by a far jump): every
    GF(2) jump matrix T^(256*2^k) the bench uses (k up to 19) is on the path, every element and the state afterwards are
    compared with the sequential host loop."""
    import torch
    rng = p3.DeviceRng(1)
    host = oracle.rng_seed_from_u64(1)
    for n in [120_000_000, 50_000_000, 17_000_001]:
        got = rng.fill_field(n)
        torch.cuda.synchronize()
        exp = oracle.rng_fill_field(host, n)
        got_h = p3.host_u32(got)
        del got
        assert np.array_equal(got_h, exp), n
        assert rng.state() == list(host), n
        del got_h, exp
    rng.close()


def _fp(p3, oracle, *t):
    return p3.FriParameters(*t), oracle.FriParams(*t)


@pytest.mark.parametrize("hash", ["poseidon2", "keccak"])
@pytest.mark.parametrize("log_n,t", [(1, (1, 0, 4, 2)), (3, (2, 2, 2, 1)), (6, (1, 0, 9, 5)), (10, (1, 0, 12, 8)), (12, (2, 1, 6, 6))])
def test_hiding_proof_bytes_equal_oracle(p3, oracle, hash, log_n, t):
    gfp, ofp = _fp(p3, oracle, *t)
    kind = oracle.HASH_KECCAK if hash == "keccak" else oracle.HASH_POSEIDON2
    pr = p3.FibAirProver(log_n, params=gfp, hash=hash, hiding=True, seed=1)
    for a, b in [(0, 1), (5, 9)]:
        proof = pr.prove(a, b)
        ref = oracle.prove_fib_air_hiding(a, b, log_n, ofp, hash=kind, seed=1)
        assert len(proof) == len(ref)
        if proof != ref:
            w1, w2 = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
            first = int(np.nonzero(w1 != w2)[0][0])
            pytest.fail("proof words differ first at %d of %d" % (first, len(w1)))
        x = oracle.fib_public_x(a, b, 1 << log_n)
        assert oracle.verify_fib_air_hiding(proof, a, b, x, log_n, ofp, hash=kind) == 0
        p3.verify_fib_air(proof, a, b, x, log_n, gfp, hash=hash, hiding=True)  # the product's own host verifier
    pr.close()


def test_reference_configuration_end_to_end(p3, oracle):
    """run_fib_air_zk as the reference wires it (fib_air.rs:27-75): Keccak hashes, hiding MMCS and PCS seeded with 1,
    create_test_fri_params(mmcs, 2), n = 8, x = 21 — prove on the device, verify on the host, report."""
    gfp, ofp = _fp(p3, oracle, 2, 2, 2, 1)
    assert p3.run_fib_air(log_n=3, params=gfp, hash="keccak", hiding=True) == "fib_air zk ok (n=8, x=21)"
    proof = p3.FibAirProver(3, params=gfp, hash="keccak", hiding=True).prove(0, 1)
    assert proof == oracle.prove_fib_air_hiding(0, 1, 3, ofp, hash=oracle.HASH_KECCAK, seed=1)
    with pytest.raises(p3.P3HipError):
        p3.verify_fib_air(proof, 0, 1, 22, 3, gfp, hash="keccak", hiding=True)


def test_hiding_seed_changes_the_proof_not_the_statement(p3, oracle):
    gfp, ofp = _fp(p3, oracle, 1, 0, 8, 4)
    p1 = p3.FibAirProver(8, params=gfp, hiding=True, seed=1).prove(0, 1)
    p2 = p3.FibAirProver(8, params=gfp, hiding=True, seed=2).prove(0, 1)
    x = oracle.fib_public_x(0, 1, 256)
    assert p1 != p2
    assert oracle.verify_fib_air_hiding(p1, 0, 1, x, 8, ofp) == 0 and oracle.verify_fib_air_hiding(p2, 0, 1, x, 8, ofp) == 0
    assert p2 == oracle.prove_fib_air_hiding(0, 1, 8, ofp, seed=2)


def test_hiding_one_fill_per_stream_equals_piecewise_fills(p3, oracle, monkeypatch):
    """The prover draws each of the three streams in ONE fill; beyond 2^30 raw draws it falls back to a fill per
    buffer (forced here through P3HIP_HIDING_PIECEWISE): same bytes either way, equal to the oracle's sequential draws."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 10, 6)
    ref = oracle.prove_fib_air_hiding(3, 4, 11, ofp, hash=oracle.HASH_KECCAK, seed=1)
    whole = p3.FibAirProver(11, params=gfp, hash="keccak", hiding=True, seed=1)
    monkeypatch.setenv("P3HIP_HIDING_PIECEWISE", "1")
    pieces = p3.FibAirProver(11, params=gfp, hash="keccak", hiding=True, seed=1)
    monkeypatch.delenv("P3HIP_HIDING_PIECEWISE")
    assert whole.prove(3, 4) == ref
    assert pieces.prove(3, 4) == ref
    whole.close(), pieces.close()


def test_hiding_batch_pool(p3, oracle):
    """The prover pool in the reference's hiding configuration: every proof equals the oracle's (the streams restart from
    the seed for every proof), through prove and through submit / collect."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 8, 4)
    pool = p3.FibAirBatchProver(7, n_provers=3, params=gfp, hash="keccak", hiding=True, seed=1)
    try:
        inst = [(i, i + 1) for i in range(7)]
        ref = [oracle.prove_fib_air_hiding(a, b, 7, ofp, hash=oracle.HASH_KECCAK, seed=1) for a, b in inst]
        assert pool.prove(inst) == ref
        t1, t2 = pool.submit(inst[:3]), pool.submit(inst[3:])
        assert pool.collect(t2) == ref[3:] and pool.collect(t1) == ref[:3]
    finally:
        pool.close()


def test_hiding_headline_size_verifies(p3, oracle):
    """2^18-row trace (randomized to 2^19, LDE 2^20), benchmark FRI parameters: the oracle's verifier accepts."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 100, 16)
    pr = p3.FibAirProver(18, params=gfp, hiding=True)
    proof = pr.prove(0, 1)
    assert oracle.verify_fib_air_hiding(proof, 0, 1, oracle.fib_public_x(0, 1, 1 << 18), 18, ofp) == 0
    assert pr.prove(0, 1) == proof
    pr.close()


def test_hiding_bench_size_in_the_reference_configuration(p3, oracle):
    """bench.py --hash keccak --hiding at its size: 2^20-row trace (randomized to 2^21, LDE 2^22: the W = 6 trace matrix and the
    chunk LDEs from coefficients go through the narrow plan at this size), Keccak hashes, seed 1.  The oracle's independent
    verifier accepts the proof for x and rejects it for x + 1; a second proof has the same bytes (the streams restart)."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 100, 16)
    pr = p3.FibAirProver(20, params=gfp, hash="keccak", hiding=True, seed=1)
    proof = pr.prove(0, 1)
    x = oracle.fib_public_x(0, 1, 1 << 20)
    assert oracle.verify_fib_air_hiding(proof, 0, 1, x, 20, ofp, hash=oracle.HASH_KECCAK) == 0
    assert oracle.verify_fib_air_hiding(proof, 0, 1, (x + 1) % 0x78000001, 20, ofp, hash=oracle.HASH_KECCAK) != 0
    assert pr.prove(0, 1) == proof
    pr.close()


def test_hiding_bench_size_proof_bytes_equal_oracle_slow(p3, oracle):
    """The bench's own instance (2^20-row trace, Keccak hashes, hiding, seed 1, benchmark FRI parameters) BYTE FOR BYTE against
    the oracle prover on all host cores (accept / reject alone is blind to the random values: the far jumps of the three
    streams, the salts of 2^22-row trees and the blinding all enter these bytes).  About a minute of CPU."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 100, 16)
    pr = p3.FibAirProver(20, params=gfp, hash="keccak", hiding=True, seed=1)
    proof = pr.prove(0, 1)
    pr.close()
    oracle.set_threads(oracle.test_threads())
    try:
        ref = oracle.prove_fib_air_hiding(0, 1, 20, ofp, hash=oracle.HASH_KECCAK, seed=1)
    finally:
        oracle.set_threads(1)
    assert len(proof) == len(ref)
    if proof != ref:
        w1, w2 = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
        pytest.fail("proof words differ first at %d of %d" % (int(np.nonzero(w1 != w2)[0][0]), len(w1)))


@pytest.mark.parametrize("hash", ["poseidon2", "keccak"])
def test_one_launch_prover_of_tiny_instances(p3, oracle, hash):
    """prover_tiny.hip.inc: under the latency profile a hiding proof whose LDE domain has at most 2^8 points (and whose random
    streams fit one wave's small fill, proof-of-work <= 4 bits) is ONE kernel launch of one workgroup — the reference's own
    instance (n = 8, x = 21, create_test_fri_params(_, 2): log_n 3, FRI (2, 2, 2, 1), fib_air.rs:56-72) and log_n 1..6 around it.
    Bytes equal the oracle prover's for both hashes, twice per prover (arena reuse), for several first rows and seeds; the
    throughput profile (the multi-launch sequence) gives the same bytes; sizes past the bound fall back to the general path."""
    kind = oracle.HASH_KECCAK if hash == "keccak" else oracle.HASH_POSEIDON2
    # (log_final_poly_len must stay below log_n + 1: 1 for log_n = 1); log_n 6 at blowup 4 = 2^9 points: general path
    cases = [(log_n, (2, 2 if log_n > 1 else 1, 2, 1)) for log_n in range(1, 7)]
    cases += [(1, (1, 0, 4, 2)), (2, (1, 1, 3, 0)), (4, (1, 0, 6, 3)), (5, (2, 1, 5, 4)), (6, (1, 3, 9, 2)), (4, (3, 0, 2, 1)), (3, (4, 2, 3, 0))]
    for log_n, t in cases:
        gfp, ofp = _fp(p3, oracle, *t)
        for seed in (1, 77):
            provers = [p3.FibAirProver(log_n, params=gfp, hash=hash, hiding=True, seed=seed, profile=pf) for pf in ("latency", "throughput")]
            for a, b in [(0, 1), (5, 9)]:
                ref = oracle.prove_fib_air_hiding(a, b, log_n, ofp, hash=kind, seed=seed)
                for pr in provers:
                    proof = pr.prove(a, b)
                    assert len(proof) == len(ref), (log_n, t, seed, pr.profile)
                    if proof != ref:
                        w1, w2 = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
                        pytest.fail("log_n %d, fri %r, seed %d, %s profile: proof words differ first at %d of %d" % (
                            log_n, t, seed, pr.profile, int(np.nonzero(w1 != w2)[0][0]), len(w1)))
                x = oracle.fib_public_x(a, b, 1 << log_n)
                assert oracle.verify_fib_air_hiding(ref, a, b, x, log_n, ofp, hash=kind) == 0
            for pr in provers:
                pr.close()
    # the reference's report entry point proves exactly this instance (Keccak, hiding, seed 1) through the latency profile
    assert p3.run_fib_air_zk_report().startswith("fib_air zk ok (n=8, x=21)")


def test_hiding_prover_at_its_largest_domain(p3, oracle):
    """The hiding prover admits LDE domains up to 2^24 points (log_n + 1 + log_blowup <= 24; prover.h MAX_LOG_DOMAIN_HIDING).  At
    the bound — 2^22-row trace, randomized to 2^23, blowup 2, Keccak hashes — the COMPLETE proof bytes equal the oracle prover's
    (16 threads: ~25 s), the oracle's verifier accepts them and rejects another public value; one past the bound is refused at creation."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 20, 8)
    pr = p3.FibAirProver(22, params=gfp, hash="keccak", hiding=True, seed=1)
    proof = pr.prove(0, 1)
    pr.close()
    x = oracle.fib_public_x(0, 1, 1 << 22)
    assert oracle.verify_fib_air_hiding(proof, 0, 1, x, 22, ofp, hash=oracle.HASH_KECCAK) == 0
    assert oracle.verify_fib_air_hiding(proof, 0, 1, (x + 1) % 0x78000001, 22, ofp, hash=oracle.HASH_KECCAK) != 0
    oracle.set_threads(oracle.test_threads())
    try:
        ref = oracle.prove_fib_air_hiding(0, 1, 22, ofp, hash=oracle.HASH_KECCAK, seed=1)
    finally:
        oracle.set_threads(1)
    assert len(proof) == len(ref)
    if proof != ref:
        w1, w2 = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
        pytest.fail("hiding proof at the largest domain: words differ first at %d of %d" % (int(np.nonzero(w1 != w2)[0][0]), len(w1)))
    with pytest.raises(p3.P3HipError):
        p3.FibAirProver(23, params=gfp, hash="keccak", hiding=True, seed=1)


def test_hiding_bench_size_poseidon2_proof_bytes_equal_oracle(p3, oracle):
    """The hiding protocol under the Poseidon2 hashes at the bench's size (2^20-row trace, benchmark FRI parameters): complete bytes
    against the oracle prover (the Keccak twin is the test above the previous one)."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 100, 16)
    pr = p3.FibAirProver(20, params=gfp, hash="poseidon2", hiding=True, seed=1)
    proof = pr.prove(0, 1)
    pr.close()
    oracle.set_threads(oracle.test_threads())
    try:
        ref = oracle.prove_fib_air_hiding(0, 1, 20, ofp, hash=oracle.HASH_POSEIDON2, seed=1)
    finally:
        oracle.set_threads(1)
    assert proof == ref


@pytest.mark.parametrize("log_n", [14, 15, 16, 17])
def test_hiding_proof_bytes_where_the_narrow_plan_takes_over(p3, oracle, log_n):
    """Byte for byte against the oracle prover at the sizes where the hiding prover's transforms switch plans: the randomized
    trace (2^(n+1) x 6) and the chunk matrices (from coefficients) enter the narrow plan at 2^16 rows, i.e. log_n = 15."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 6, 4)
    pr = p3.FibAirProver(log_n, params=gfp, hash="keccak", hiding=True, seed=1)
    assert pr.prove(3, 4) == oracle.prove_fib_air_hiding(3, 4, log_n, ofp, hash=oracle.HASH_KECCAK, seed=1)
    pr.close()


@pytest.mark.parametrize("hash", ["poseidon2", "keccak"])
def test_hiding_mmcs_commit_and_openings(p3, oracle, hash):
    """MerkleTreeHidingMmcs on the device: salts drawn from the MMCS's own stream in input order, leaves m0 || s0 || m1 || s1;
    root and openings equal the oracle tree over the same interleaved (matrix, salt) list; two commits continue the stream."""
    rng = np.random.default_rng(9)
    kind = oracle.HASH_KECCAK if hash == "keccak" else oracle.HASH_POSEIDON2
    mmcs = p3.MerkleTreeHidingMmcs(hash, seed=1)
    host = oracle.rng_seed_from_u64(1)
    for mats_shape in ([(64, 6)], [(128, 4), (128, 4), (32, 3)]):
        mats = [rng.integers(0, 0x78000001, size=s, dtype=np.uint64).astype(np.uint32) for s in mats_shape]
        root, tree = mmcs.commit(mats)
        inter = []
        for m in mats:
            inter += [m, oracle.rng_fill_field(host, m.shape[0] * 4).reshape(m.shape[0], 4)]
        exp_root, otree = oracle.mmcs_commit(inter, kind=kind)
        assert np.array_equal(root, exp_root)
        for index in (0, 5, mats[0].shape[0] - 1):
            vals, (salts, path) = mmcs.open_batch(index, tree)
            orows, opath = otree.open_batch(index)
            got = np.concatenate([np.concatenate([v, s]) for v, s in zip(vals, salts)])
            assert np.array_equal(got, orows) and np.array_equal(path, opath)
            dims = [(m.shape[0], m.shape[1]) for m in inter]
            assert oracle.mmcs_verify_batch(root, dims, index, got, path, kind=kind)
        tree.free()


def test_hiding_provers_concurrently_keep_their_bytes(p3, oracle):
    """Four hiding provers on four host threads (each with its main stream and its fill side stream) prove the same three instances
    over and over: every proof equals the oracle's.  A missed cross-stream dependency — a fill racing its consumer — would differ."""
    import threading
    import torch
    gfp, ofp = _fp(p3, oracle, 1, 0, 12, 6)
    log_n, insts = 13, [(0, 1), (3, 4), (9, 2)]
    ref = {ab: oracle.prove_fib_air_hiding(ab[0], ab[1], log_n, ofp, hash=oracle.HASH_KECCAK, seed=1) for ab in insts}
    bad = []

    def worker(k):
        torch.cuda.set_device(0)
        pr = p3.FibAirProver(log_n, params=gfp, hash="keccak", hiding=True, seed=1)
        for r in range(18):
            ab = insts[(r + k) % 3]
            if pr.prove(*ab) != ref[ab]:
                bad.append((k, r, ab))
        pr.close()
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad[:5]
