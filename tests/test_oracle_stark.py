"""CPU tests of the oracle's fib_air prover/verifier pair (oracle/stark.c).  Upstream parity of the
protocol glue is unpinned (no fixture exists in the reference); these tests pin self-consistency:
an independently written verifier accepts the prover's output and rejects perturbed statements."""
import numpy as np
import pytest


@pytest.mark.parametrize("log_n", [1, 2, 3, 6, 9])
def test_prove_then_verify(oracle, log_n):
    fp = oracle.FriParams(1, 0, 12, 6)
    proof = oracle.prove_fib_air(0, 1, log_n, fp)
    x = oracle.fib_public_x(0, 1, 1 << log_n)
    assert oracle.verify_fib_air(proof, 0, 1, x, log_n, fp) == 0
    assert oracle.prove_fib_air(0, 1, log_n, fp) == proof  # deterministic
    assert oracle.verify_fib_air(proof, 0, 1, x + 1, log_n, fp) != 0   # wrong public value
    assert oracle.verify_fib_air(proof, 1, 1, x, log_n, fp) != 0       # wrong first row


def test_reference_instance_n8_x21(oracle):
    # the reference proves n = 8, x = 21 (native/src/fib_air.rs:56-57,68)
    fp = oracle.FriParams(1, 0, 10, 4)
    proof = oracle.prove_fib_air(0, 1, 3, fp)
    assert oracle.verify_fib_air(proof, 0, 1, 21, 3, fp) == 0
    assert oracle.verify_fib_air(proof, 0, 1, 22, 3, fp) == 10  # OodEvaluationMismatch


@pytest.mark.parametrize("fp", [(1, 0, 8, 0), (2, 0, 5, 3), (2, 2, 5, 4), (1, 3, 7, 8), (3, 1, 4, 2)])
def test_fri_parameter_variants(oracle, fp):
    fp = oracle.FriParams(*fp)
    proof = oracle.prove_fib_air(5, 8, 7, fp)
    assert oracle.verify_fib_air(proof, 5, 8, oracle.fib_public_x(5, 8, 128), 7, fp) == 0


def test_every_tampered_word_is_rejected(oracle):
    fp = oracle.FriParams(1, 0, 3, 4)
    proof = oracle.prove_fib_air(0, 1, 4, fp)
    x = oracle.fib_public_x(0, 1, 16)
    words = np.frombuffer(proof, dtype=np.uint32)
    rng = np.random.default_rng(0)
    for pos in rng.choice(len(words), size=60, replace=False):
        bad = words.copy()
        bad[pos] = (int(bad[pos]) + 1) % 0x78000001
        assert oracle.verify_fib_air(bad.tobytes(), 0, 1, x, 4, fp) != 0, pos


def test_benchmark_parameters_small(oracle):
    fp = oracle.FriParams()  # log_blowup 1, final poly len 1, 100 queries, 16 PoW bits
    proof = oracle.prove_fib_air(0, 1, 10, fp)
    assert oracle.verify_fib_air(proof, 0, 1, oracle.fib_public_x(0, 1, 1024), 10, fp) == 0


def test_oracle_under_address_and_ub_sanitizers():
    """The whole C restatement (both hash configurations, trees, transforms, every truncation of a proof) under
    ASan + UBSan on the CPU: `make -C oracle sanitize` (GPU sanitizers are not available on the pool)."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "cannot find -lasan" in r.stderr + r.stdout:
        pytest.skip("libasan not installed")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "oracle selftest ok" in r.stdout
