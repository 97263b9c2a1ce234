"""CPU tests of the host logic and of the C-ABI library surface (no compute calls: no GPU here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(p3):
    hdr = open(os.path.join(ROOT, "include", "p3hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # prototypes only, not names mentioned in comments
    declared = sorted(set(re.findall(r"\b(p3hip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    lib = C.CDLL(p3._lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert set(p3._lib.declared_symbols()) == set(declared)


def test_backend_selector_mirrors_reference(p3):
    # gpu_dft.rs:53-63: case-insensitive names, unknown -> Err("unknown backend '...'")
    try:
        for name, kind in [("cpu", 0), ("VULKAN", 1), ("Metal", 2), ("webgpu", 3), ("hip", 4)]:
            p3.set_backend_kind_from_str(name)
            assert int(p3.get_backend_kind()) == kind
        with pytest.raises(ValueError, match="unknown backend 'cuda'"):
            p3.set_backend_kind_from_str("cuda")
        assert int(p3.get_backend_kind()) == 4  # unchanged by the failed call
        assert p3.take_last_error() is None  # the failed call's message was taken by the ValueError
    finally:
        p3.set_backend_kind_from_str("hip")
    assert p3.GpuDft().backend == p3.BackendKind.Hip  # Default reads the global (gpu_dft.rs:76-83)


def test_no_gpu_is_an_error_not_a_fallback(p3):
    import numpy as np
    ok, msg = p3.is_available()
    if ok:
        pytest.skip("GPU present")
    assert msg.startswith("HIP unavailable")
    with pytest.raises(p3.P3HipError):
        p3.GpuDft.with_backend(p3.BackendKind.Hip).dft_batch(np.zeros((4, 2), np.uint32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "plonky3-mobile_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hip.h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("test oracle", ""), os.path.join(dirpath, f)


def test_product_verifier_accepts_oracle_proofs_and_rejects_tampering(p3, oracle):
    """p3hip_verify_fib_air is host code (no GPU needed): cross-check it against the oracle's prover, and the
    oracle's verifier against the same inputs."""
    import numpy as np
    for log_n, t in [(3, (1, 0, 10, 4)), (7, (2, 2, 6, 5)), (10, (1, 0, 100, 16))]:
        ofp, gfp = oracle.FriParams(*t), p3.FriParameters(*t)
        proof = oracle.prove_fib_air(0, 1, log_n, ofp)
        x = p3.fib_public_x(0, 1, 1 << log_n)
        assert x == oracle.fib_public_x(0, 1, 1 << log_n)
        p3.verify_fib_air(proof, 0, 1, x, log_n, gfp)  # accepts
        with pytest.raises(p3.P3HipError, match="OodEvaluationMismatch"):
            p3.verify_fib_air(proof, 0, 1, x + 1, log_n, gfp)
        words = np.frombuffer(proof, dtype=np.uint32)
        rng = np.random.default_rng(log_n)
        for pos in rng.choice(len(words), size=25, replace=False):
            bad = words.copy()
            bad[pos] = (int(bad[pos]) + 1) % 0x78000001
            with pytest.raises(p3.P3HipError):
                p3.verify_fib_air(bad.tobytes(), 0, 1, x, log_n, gfp)
            assert oracle.verify_fib_air(bad.tobytes(), 0, 1, x, log_n, ofp) != 0


def test_product_verifier_keccak_configuration(p3, oracle):
    """The host verifier under the reference's own hashes (Keccak MMCS + SerializingChallenger32 / Keccak-256
    HashChallenger) against the oracle's prover for the same configuration; the two were written separately."""
    import numpy as np
    K = oracle.HASH_KECCAK
    for log_n, t in [(3, (1, 0, 10, 4)), (6, (2, 2, 6, 5)), (9, (1, 1, 20, 8))]:
        ofp, gfp = oracle.FriParams(*t), p3.FriParameters(*t)
        proof = oracle.prove_fib_air(3, 4, log_n, ofp, hash=K)
        x = p3.fib_public_x(3, 4, 1 << log_n)
        p3.verify_fib_air(proof, 3, 4, x, log_n, gfp, hash="keccak")  # accepts
        with pytest.raises(p3.P3HipError):
            p3.verify_fib_air(proof, 3, 4, x, log_n, gfp)              # Poseidon2 verifier, Keccak proof
        with pytest.raises(p3.P3HipError, match="OodEvaluationMismatch"):
            p3.verify_fib_air(proof, 3, 4, x + 1, log_n, gfp, hash="keccak")
        words = np.frombuffer(proof, dtype=np.uint32)
        rng = np.random.default_rng(log_n)
        for pos in rng.choice(len(words), size=25, replace=False):
            bad = words.copy()
            bad[pos] = (int(bad[pos]) + 1) % 0x78000001
            with pytest.raises(p3.P3HipError):
                p3.verify_fib_air(bad.tobytes(), 3, 4, x, log_n, gfp, hash="keccak")
    with pytest.raises(ValueError):
        p3.verify_fib_air(proof, 3, 4, x, log_n, gfp, hash="sha2")


def test_plan_helpers_mirror_reference_layouts(p3, oracle):
    """FftStageParams / prepare_compute_plan / twiddle_table / bit-reversed rows (backend_vulkan.rs:784-1026): host
    integer code, checked against the oracle's statement of the same layouts."""
    import numpy as np
    pl = p3.plan
    prm = pl.params_for_stage(8, 1024, 3, 10, 7)
    assert (prm.width, prm.height, prm.stage, prm.log_n, prm.twiddle_base) == (8, 1024, 3, 10, 7)
    # ComputePlan.dispatch is the library's own launch plan (p3hip_dft_plan_bb31, host-only): stages per LDS-tiled pass.  The
    # reference dispatches one stage at a time (14 launches for 2^14 rows, 20 for 2^20: backend_vulkan.rs:1182-1294)
    plan = pl.prepare_compute_plan(128, 16384, 0, 14)
    assert plan.params.twiddle_base == 1 and sum(plan.dispatch) == 14 and 1 <= len(plan.dispatch) <= 2
    big = pl.prepare_compute_plan(2, 1 << 20, 0, 20).dispatch
    assert sum(big) == 20 and len(big) <= 3 and max(big) <= 11
    assert pl.prepare_compute_plan(3, 1, 0, 0).dispatch == () and pl.launch_plan(256, 8) == (8,)
    with pytest.raises(p3.P3HipError):
        pl.launch_plan(24, 2)  # power-of-two gate, as the DFT itself (backend_vulkan.rs:1992-1995)
    for log_n in (0, 1, 4, 9):
        tab = pl.twiddle_table(log_n)
        assert tab.size == (1 << log_n) - 1 if log_n else tab.size == 0
        if log_n:
            L = oracle.lib()
            exp = np.zeros((1 << log_n) - 1, dtype=np.uint32)
            L.p3o_twiddle_table(log_n, exp.ctypes.data_as(oracle._u32p))
            assert np.array_equal(tab, exp)
            assert all(np.array_equal(tab[(1 << s) - 1:(1 << (s + 1)) - 1], pl.twiddles_for_stage(log_n, s)) for s in range(log_n))
    assert pl.two_adic_generator(27) == 0x1a427a41
    rng = np.random.default_rng(2)
    x = rng.integers(0, 0x78000001, (64, 3), dtype=np.uint32)
    assert np.array_equal(pl.write_bit_reversed_rows_u32(x.reshape(-1), 3).reshape(64, 3), oracle.bit_reverse_rows(x))
    assert np.array_equal(pl.write_bit_reversed_rows_u32(x[:24].reshape(-1), 3), x[:24].reshape(-1))  # 24 rows: copied
    assert [pl.reverse_bits_len(i, 3) for i in range(8)] == [0, 4, 2, 6, 1, 5, 3, 7]


def test_header_is_plain_c(tmp_path):
    """include/p3hip.h is the drop-in boundary: it must compile as C11 (no C++, no torch, no HIP types)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    src = tmp_path / "hdr.c"
    src.write_text('#include "p3hip.h"\nint main(void) { return p3hip_get_backend() < -100; }\n')
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-fsyntax-only", str(src)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_product_hiding_verifier_accepts_oracle_proofs_and_rejects_tampering(p3, oracle):
    """p3hip_verify_fib_air_hiding is host code: cross-check it against the oracle's hiding prover (both hash
    configurations) and against tampered proofs."""
    import numpy as np
    for hash, kind in (("poseidon2", 0), ("keccak", 1)):
        for log_n, t in [(3, (2, 2, 2, 1)), (6, (1, 0, 7, 4))]:
            gfp, ofp = p3.FriParameters(*t), oracle.FriParams(*t)
            proof = oracle.prove_fib_air_hiding(2, 3, log_n, ofp, hash=kind)
            x = oracle.fib_public_x(2, 3, 1 << log_n)
            p3.verify_fib_air(proof, 2, 3, x, log_n, gfp, hash=hash, hiding=True)
            with pytest.raises(p3.P3HipError):
                p3.verify_fib_air(proof, 2, 3, x + 1, log_n, gfp, hash=hash, hiding=True)
            words = np.frombuffer(proof, np.uint32)
            rng = np.random.default_rng(log_n)
            for pos in rng.choice(len(words), size=25, replace=False):
                bad = words.copy()
                bad[pos] = (int(bad[pos]) + 1) % 0x78000001
                with pytest.raises(p3.P3HipError):
                    p3.verify_fib_air(bad.tobytes(), 2, 3, x, log_n, gfp, hash=hash, hiding=True)


def test_report_entry_points_honour_the_selector_without_a_gpu(p3):
    """p3hip_run_fib_air_zk stands for fib_air::run_fib_air_zk behind lib.rs:37-83: a String, never a status.  With a backend
    other than hip selected it runs nothing and says so (the reference hard-codes Vulkan at fib_air.rs:60; the drop-in does not
    override the selector); without a GPU the hip arm reports a failure as text."""
    p3.set_backend_kind_from_str("cpu")
    try:
        text = p3.run_fib_air_zk_report()
        assert text.startswith("fib_air zk failed: backend 'cpu' is selected")
    finally:
        p3.set_backend_kind_from_str("hip")
    import torch
    if not torch.cuda.is_available():
        assert p3.run_fib_air_zk_report().startswith("fib_air zk failed: ")
        assert p3.run_dft_benchmark_report().startswith("dft benchmark failed: HIP unavailable")
        p3.take_last_error()
