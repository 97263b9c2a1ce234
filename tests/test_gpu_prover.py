"""GPU parity tests of the fib_air prover: proof bytes must equal the CPU oracle's byte for byte (every
commitment, opened value, FRI layer root, witness and opening is in there), and the oracle's
independently written verifier must accept them."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fp(p3, oracle, *t):
    return p3.FriParameters(*t), oracle.FriParams(*t)


@pytest.mark.parametrize("log_n", [1, 2, 3, 5, 8, 10, 12, 14])
def test_proof_bytes_equal_oracle(p3, oracle, log_n):
    gfp, ofp = _fp(p3, oracle, 1, 0, 20, 8)
    pr = p3.FibAirProver(log_n, params=gfp)
    for a, b in [(0, 1), (7, 11)]:
        proof = pr.prove(a, b)
        ref = oracle.prove_fib_air(a, b, log_n, ofp)
        assert len(proof) == len(ref)
        if proof != ref:
            w1, w2 = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
            first = int(np.nonzero(w1 != w2)[0][0])
            pytest.fail("proof words differ first at %d of %d" % (first, len(w1)))
        assert oracle.verify_fib_air(proof, a, b, oracle.fib_public_x(a, b, 1 << log_n), log_n, ofp) == 0
    pr.close()


@pytest.mark.parametrize("t", [(1, 0, 100, 16), (2, 0, 10, 4), (2, 2, 6, 5), (1, 3, 9, 0), (3, 1, 4, 10), (1, 0, 0, 0),
                               (1, 8, 3, 2), (4, 0, 2, 1)])
def test_fri_parameter_variants(p3, oracle, t):
    gfp, ofp = _fp(p3, oracle, *t)
    pr = p3.FibAirProver(9, params=gfp)
    proof = pr.prove(3, 5)
    assert proof == oracle.prove_fib_air(3, 5, 9, ofp)
    assert oracle.verify_fib_air(proof, 3, 5, oracle.fib_public_x(3, 5, 512), 9, ofp) == 0


def test_reference_instance(p3, oracle):
    # the reference's own instance: n = 8, x = 21 (native/src/fib_air.rs:56-57)
    gfp, ofp = _fp(p3, oracle, 1, 0, 10, 4)
    proof = p3.FibAirProver(3, params=gfp).prove(0, 1)
    assert oracle.verify_fib_air(proof, 0, 1, 21, 3, ofp) == 0
    assert oracle.verify_fib_air(proof, 0, 1, 22, 3, ofp) != 0


def _assert_same_proof(proof, ref, what):
    assert len(proof) == len(ref), (what, len(proof), len(ref))
    if proof != ref:
        w1, w2 = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
        first = int(np.nonzero(w1 != w2)[0][0])
        pytest.fail("%s: proof words differ first at %d of %d" % (what, first, len(w1)))


def test_headline_2_20_proof_bytes_equal_oracle(p3, oracle):
    """BASELINE cfg2 at its own size (2^20 rows, blowup 2, the benchmark's FRI parameters: 100 queries, 16 proof-of-work bits):
    the COMPLETE proof bytes equal the oracle prover's (run on every host core: field arithmetic is exact, the bytes do not
    depend on the thread count — tests/test_oracle_stark.py), the oracle verifier accepts them, arena reuse is clean, the
    trace commitment inside the proof is the oracle's commitment of the same trace.  Until round 5 this comparison lived
    only in bench.py's cpu_baseline leg."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 100, 16)
    pr = p3.FibAirProver(20, params=gfp)
    proof = pr.prove(0, 1)
    x = oracle.fib_public_x(0, 1, 1 << 20)
    oracle.set_threads(oracle.test_threads())
    try:
        ref = oracle.prove_fib_air(0, 1, 20, ofp)
        lde = oracle.coset_lde_batch(oracle.generate_trace_rows(0, 1, 1 << 20), 1, p3.GENERATOR_MONTY, True)
        root, _ = oracle.mmcs_commit([lde])
    finally:
        oracle.set_threads(1)
    _assert_same_proof(proof, ref, "cfg2 poseidon2")
    assert oracle.verify_fib_air(proof, 0, 1, x, 20, ofp) == 0
    assert pr.prove(0, 1) == proof  # arena reuse is clean
    words = np.frombuffer(proof, np.uint32)
    assert np.array_equal(words[3:11], root)
    other = pr.prove(1, 2)
    assert other != proof and oracle.verify_fib_air(other, 1, 2, oracle.fib_public_x(1, 2, 1 << 20), 20, ofp) == 0
    pr.close()


def test_cfg4_instances_at_their_size_through_the_pool(p3, oracle):
    """BASELINE configs[3]: 64 independent 2^20-row instances with first rows (i, i + 1).  A pool of four provers (the throughput
    profile, as bench.py's ranks run them) proves a batch holding instances 5, 63, 0 and 31 at that size with the bench's FRI
    parameters; the first two are compared byte for byte with the oracle prover (all host cores), all four are accepted by the
    oracle verifier for their own public values and rejected for a neighbour's."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 100, 16)
    inst = [(5, 6), (63, 64), (0, 1), (31, 32)]
    pool = p3.FibAirBatchProver(20, n_provers=4, params=gfp)
    try:
        proofs = pool.prove(inst)
    finally:
        pool.close()
    oracle.set_threads(oracle.test_threads())
    try:
        for (a, b), pf in zip(inst[:2], proofs[:2]):
            _assert_same_proof(pf, oracle.prove_fib_air(a, b, 20, ofp), "cfg4 instance (%d, %d)" % (a, b))
    finally:
        oracle.set_threads(1)
    for (a, b), pf in zip(inst, proofs):
        x = oracle.fib_public_x(a, b, 1 << 20)
        assert oracle.verify_fib_air(pf, a, b, x, 20, ofp) == 0, (a, b)
        assert oracle.verify_fib_air(pf, a + 1, b + 1, x, 20, ofp) != 0, (a, b)


def test_concurrent_provers_on_threads(p3, oracle):
    """One prover per host thread (per-thread context + own stream), as the batch bench runs them."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 16, 6)
    out = {}

    def work(i):
        pr = p3.FibAirProver(11, params=gfp)
        out[i] = [pr.prove(i, i + 1) for _ in range(2)]
        pr.close()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    for i in range(4):
        ref = oracle.prove_fib_air(i, i + 1, 11, ofp)
        assert out[i][0] == ref and out[i][1] == ref


def test_bench_job_overlapping_steps(p3, oracle):
    """bench.py's step pipeline: step k + 1 is dealt to the prover threads before step k retires; every step still
    returns exactly its own proofs (bytes equal to the oracle's), in order, and the sink sees each proof once."""
    from plonky3_mobile_amd import bench_support as bs
    job = bs.FibAirJob(p3, 9, 1, batch=5, threads=3)
    ofp = oracle.FriParams(*[getattr(job.params, k) for k in ("log_blowup", "log_final_poly_len", "num_queries", "proof_of_work_bits")])
    try:
        seen = []
        job.step_begin()                                              # default instances (first + i)
        job.step_begin([(7, 40), (3, 41), (9, 42)], lambda i, pf: seen.append((i, pf)))
        job.step_begin([(0, 50)])
        first = job.step_end()
        second = job.step_end()
        third = job.step_end()
        assert first == [oracle.prove_fib_air(i, i + 1, 9, ofp) for i in range(5)]
        assert second == {7: oracle.prove_fib_air(40, 41, 9, ofp), 3: oracle.prove_fib_air(41, 42, 9, ofp),
                          9: oracle.prove_fib_air(42, 43, 9, ofp)}
        assert sorted(seen) == sorted(second.items())
        assert third == {0: oracle.prove_fib_air(50, 51, 9, ofp)}
    finally:
        job.close()


def test_batch_pool_submit_collect(p3, oracle):
    """The pool's pipelined form: three batches submitted before the first is collected, collected out of order; every
    proof equals the oracle's; a ninth batch in flight is refused; an unknown ticket is an error."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 12, 5)
    pool = p3.FibAirBatchProver(9, n_provers=3, params=gfp)
    try:
        batches = [[(10 * k + i, 10 * k + i + 1) for i in range(4 + k)] for k in range(3)]
        tickets = [pool.submit(bt) for bt in batches]
        for k in (1, 0, 2):
            assert pool.collect(tickets[k]) == [oracle.prove_fib_air(a, b, 9, ofp) for a, b in batches[k]]
        assert pool.prove(batches[0][:2]) == [oracle.prove_fib_air(a, b, 9, ofp) for a, b in batches[0][:2]]
        with pytest.raises(p3.P3HipError):
            pool.collect(tickets[0])  # already collected
        more = [pool.submit([(0, 1)]) for _ in range(8)]
        with pytest.raises(p3.P3HipError):
            pool.submit([(0, 1)])
        for t in more:
            assert pool.collect(t) == [oracle.prove_fib_air(0, 1, 9, ofp)]
        assert pool.collect(pool.submit([])) == []
    finally:
        pool.close()


@pytest.mark.parametrize("hash", ["poseidon2", "keccak"])
def test_enqueue_finish_two_proofs_in_flight(p3, oracle, hash):
    """enqueue / finish: the second proof's launches queue behind the first on the prover's stream; results come back in
    order, byte for byte; a third enqueue and a finish with nothing in flight are refused; prove() still works after."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 14, 6)
    kind = oracle.HASH_KECCAK if hash == "keccak" else oracle.HASH_POSEIDON2
    pr = p3.FibAirProver(10, params=gfp, hash=hash)
    ref = [oracle.prove_fib_air(a, a + 1, 10, ofp, hash=kind) for a in range(5)]
    pr.enqueue(0, 1)
    pr.enqueue(1, 2)
    with pytest.raises(p3.P3HipError):
        pr.enqueue(2, 3)
    with pytest.raises(p3.P3HipError):
        pr.prove(2, 3)  # synchronous prove with proofs in flight
    assert pr.finish() == ref[0]
    pr.enqueue(2, 3)
    assert pr.finish() == ref[1]
    pr.enqueue(3, 4)
    assert pr.finish() == ref[2]
    assert pr.finish() == ref[3]
    with pytest.raises(p3.P3HipError):
        pr.finish()
    assert pr.prove(4, 5) == ref[4]
    pr.close()


def test_bad_parameters(p3, oracle):
    with pytest.raises(p3.P3HipError):
        p3.FibAirProver(0)
    # upstream asserts log_min_height > log_final_poly_len + log_blowup when log_final_poly_len > 0
    with pytest.raises(p3.P3HipError):
        p3.FibAirProver(9, params=p3.FriParameters(1, 9, 3, 2))
    with pytest.raises(ValueError):
        oracle.prove_fib_air(0, 1, 9, oracle.FriParams(1, 9, 3, 2))
    with pytest.raises(p3.P3HipError):
        p3.FibAirProver(27, params=p3.FriParameters(log_blowup=2))
    with pytest.raises(p3.P3HipError):
        p3.FibAirProver(26, params=p3.FriParameters(log_blowup=1))  # 2^27-point domain: above the tested bound (2^26)


def test_cfg3_2_24_blowup4_proof_bytes_equal_oracle(p3, oracle):
    """BASELINE configs[2] at its own size AND with bench.py's parameters (2^24-row trace, blowup 4, 100 queries, 16
    proof-of-work bits — "FRI-fold-heavy"): the complete proof bytes equal the oracle prover's.  Every commitment is in those
    bytes: the trace root over the 2^26-leaf tree, the quotient root (quotient at blowup 4), the 24 FRI layer roots (the
    2^26-point folds), plus opened values, witness and all 100 query openings.  The oracle proof runs on every host core
    (~1-3 min on the GPU box's 16; ~15 GB of host memory) — the single most expensive test of the suite, which is why the
    smaller parameter set (50 queries, 8 bits) it replaced was accept / reject only.  Also: the oracle verifier accepts, and
    rejects another public value."""
    gfp, ofp = _fp(p3, oracle, 2, 0, 100, 16)
    pr = p3.FibAirProver(24, params=gfp)
    proof = pr.prove(0, 1)
    pr.close()
    x = oracle.fib_public_x(0, 1, 1 << 24)
    assert oracle.verify_fib_air(proof, 0, 1, x, 24, ofp) == 0
    assert oracle.verify_fib_air(proof, 0, 1, x + 1, 24, ofp) != 0
    oracle.set_threads(oracle.test_threads())
    try:
        ref = oracle.prove_fib_air(0, 1, 24, ofp)
    finally:
        oracle.set_threads(1)
    words, rwords = np.frombuffer(proof, np.uint32), np.frombuffer(ref, np.uint32)
    assert np.array_equal(words[3:11], rwords[3:11]), "cfg3: trace commitment differs from the oracle's"
    assert np.array_equal(words[11:19], rwords[11:19]), "cfg3: quotient commitment differs from the oracle's"
    _assert_same_proof(proof, ref, "cfg3")


def test_largest_domain_the_prover_admits(p3, oracle):
    """The prover admits LDE domains up to 2^26 points (log_n + log_blowup <= 26: BASELINE configs[2]'s size, the largest
    any test or bench has run; until round 5 the bound was the field's two-adicity, 27, which nothing had ever exercised).
    At the bound with the OTHER split (2^25 rows, blowup 2): the oracle verifier accepts; one past it is refused at creation."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 30, 8)
    pr = p3.FibAirProver(25, params=gfp)
    proof = pr.prove(2, 5)
    pr.close()
    x = oracle.fib_public_x(2, 5, 1 << 25)
    assert oracle.verify_fib_air(proof, 2, 5, x, 25, ofp) == 0
    assert oracle.verify_fib_air(proof, 2, 5, x + 1, 25, ofp) != 0
    with pytest.raises(p3.P3HipError):
        p3.FibAirProver(26, params=gfp)
    with pytest.raises(p3.P3HipError):
        p3.FibAirProver(25, params=p3.FriParameters(2, 0, 30, 8))


def test_dft_benchmark_harness(p3, oracle):
    """run_dft_benchmark (fib_air.rs:98-222) incl. the reference's equality check against the CPU path."""
    text, rows = p3.run_dft_benchmark(cases=[(256, 8), (4096, 32), (256, 1000)], repeats=3, cpu_dft=oracle.dft_batch)
    assert text.startswith("dft benchmark (repeats=3, warmup=1, stats=avg/median/p95)")
    assert len(rows) == 3 and all(r["hip_kernel"][0] > 0 for r in rows)
    assert p3.percentile_ms([3.0, 1.0, 2.0, 4.0], 0.95) == 4.0 and p3.percentile_ms([], 0.5) == 0.0


def test_run_fib_air_mirrors_reference_report(p3):
    # run_fib_air_zk returns "fib_air zk ok (n=8, x=21)" (fib_air.rs:74); ours proves + verifies on the hip backend
    assert p3.run_fib_air(params=p3.FriParameters(1, 0, 10, 4)) == "fib_air ok (n=8, x=21)"
    assert p3.run_fib_air(log_n=12).startswith("fib_air ok (n=4096, x=")


def test_batch_prover_pool(p3, oracle):
    """p3hip_fib_batch_*: a pool of provers inside the library; results in instance order, repeatable."""
    gfp, ofp = _fp(p3, oracle, 1, 0, 16, 6)
    pool = p3.FibAirBatchProver(10, n_provers=4, params=gfp)
    inst = [(i, 2 * i + 1) for i in range(11)]
    proofs = pool.prove(inst)
    assert [len(p) > 1000 for p in proofs] == [True] * 11
    for (a, b), pf in zip(inst[:4], proofs[:4]):
        assert pf == oracle.prove_fib_air(a, b, 10, ofp)
    for (a, b), pf in zip(inst, proofs):
        p3.verify_fib_air(pf, a, b, p3.fib_public_x(a, b, 1 << 10), 10, gfp)
    assert pool.prove(inst[:3]) == proofs[:3]
    assert pool.prove([]) == []
    pool.close()


def test_reference_fri_parameters_n8(p3, oracle):
    """The reference's own call: n = 8, x = 21, create_test_fri_params(challenge_mmcs, 2) (fib_air.rs:56-62) =
    log_blowup 2, log_final_poly_len 2, 2 queries, 1 proof-of-work bit [UPSTREAM-RECALL for the values]."""
    gfp, ofp = _fp(p3, oracle, 2, 2, 2, 1)
    assert p3.run_fib_air(log_n=3, params=gfp) == "fib_air ok (n=8, x=21)"
    proof = p3.FibAirProver(3, params=gfp).prove(0, 1)
    assert proof == oracle.prove_fib_air(0, 1, 3, ofp)
    assert oracle.verify_fib_air(proof, 0, 1, 21, 3, ofp) == 0


@pytest.mark.parametrize("hash", ["poseidon2", "keccak"])
def test_grind_search_continues_after_an_empty_first_range(p3, oracle, hash, monkeypatch):
    """The device transcript covers 16x the expected number of proof-of-work candidates in its first launch; when that
    range holds no witness the host continues the search range by range and redoes the query phase.  Forced here by a
    first range of 256 candidates against 12 proof-of-work bits: the proof must still equal the oracle's."""
    monkeypatch.setenv("P3HIP_GRIND_FIRST_LOG", "8")
    gfp, ofp = _fp(p3, oracle, 1, 0, 12, 12)
    kind = oracle.HASH_KECCAK if hash == "keccak" else oracle.HASH_POSEIDON2
    pr = p3.FibAirProver(9, params=gfp, hash=hash)
    hit_continuation = False
    for a in range(4):
        proof = pr.prove(a, a + 1)
        assert proof == oracle.prove_fib_air(a, a + 1, 9, ofp, hash=kind)
        witness = int(np.frombuffer(proof[-4:], np.uint32)[0])
        hit_continuation |= int(oracle.from_monty(np.array([witness]))[0]) >= 256
    assert hit_continuation, "no instance needed the continuation path: pick other instances"
    # Round 2's GPU fault lived here: on a miss ts_queries_kernel returned before it had written the query indices, and
    # query_gather_kernel — queued behind it unconditionally — indexed the trees with whatever the arena held.  Pinned: a
    # proof that DID sample indices ran first (a = 0 ... leave non-zero indices in the buffer), and after every later miss
    # the buffer the gather read is all zero.
    misses, idx = pr.grind_miss_probe()
    assert misses >= 1 and len(idx) == gfp.num_queries and not idx.any(), (misses, idx)
    # the same with two proofs in flight: when the older one needs the continuation its arena has been reused by the
    # newer one, so finish() runs it again from the start before continuing the search
    pr.enqueue(0, 1)
    for a in range(1, 4):
        pr.enqueue(a, a + 1)
        assert pr.finish() == oracle.prove_fib_air(a - 1, a, 9, ofp, hash=kind)
    assert pr.finish() == oracle.prove_fib_air(3, 4, 9, ofp, hash=kind)
    pr.close()


def test_prove_into_writes_the_same_bytes(p3, oracle):
    """p3hip_fib_prover_prove_into: the proof written into a caller's buffer (bench.py: the pinned staging row of the step's gather)
    equals the bytes prove() returns; a buffer that is too small is refused with the needed size, nothing is truncated."""
    import ctypes as C
    import torch
    gfp, ofp = _fp(p3, oracle, 1, 0, 10, 6)
    pr = p3.FibAirProver(11, params=gfp)
    ref = pr.prove(5, 6)
    buf = torch.zeros(len(ref) + 64, dtype=torch.uint8).pin_memory()
    n = pr.prove_into(5, 6, buf.data_ptr(), buf.numel())
    assert n == len(ref) and bytes(buf[:n].numpy()) == ref == oracle.prove_fib_air(5, 6, 11, ofp)
    assert not buf[n:].any()
    with pytest.raises(p3.P3HipError, match="needs %d bytes" % len(ref)):
        pr.prove_into(5, 6, buf.data_ptr(), 100)
    pr.close()
