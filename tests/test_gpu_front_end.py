"""The reference's two report-returning entry points (native/src/lib.rs:37-131 -> fib_air::run_fib_air_zk / run_dft_benchmark)
as C entries of libp3hip, called through ctypes the way the patched native/src/fib_air.rs calls them."""
import re

import pytest

pytestmark = pytest.mark.gpu


def test_run_fib_air_zk_reports_the_reference_instance(p3):
    # fib_air.rs:74: Ok(format!("fib_air zk ok (n={n}, x={x})")) with n = 8, x = 21 — the reference's own configuration
    # (Keccak hashes, hiding MMCS + PCS, seed 1), proved on the device, verified by the host verifier
    assert p3.run_fib_air_zk_report() == "fib_air zk ok (n=8, x=21)"
    assert p3.take_last_error() is None


def test_run_fib_air_zk_honours_the_selector(p3):
    p3.set_backend_kind_from_str("vulkan")
    try:
        text = p3.run_fib_air_zk_report()
    finally:
        p3.set_backend_kind_from_str("hip")
    assert "failed" in text and "vulkan" in text
    assert p3.run_fib_air_zk_report() == "fib_air zk ok (n=8, x=21)"


def test_run_dft_benchmark_report_with_the_oracle_as_cpu_column(p3, oracle):
    text = p3.run_dft_benchmark_report(cpu_dft=oracle.dft_batch)
    lines = text.splitlines()
    assert lines[0] == "dft benchmark (repeats=10, warmup=1, stats=avg/median/p95)", text[:300]
    assert len(lines) == 12 and "failed" not in text  # the reference's 11 shapes; equality with the CPU column checked inside
    shapes = [tuple(int(v) for v in re.match(r"h=(\d+), w=(\d+):", l).groups()) for l in lines[1:]]
    assert shapes == list(p3.BENCHMARK_CASES)
    for l in lines[1:]:
        for col in ("cpu(avg=", "hip_e2e(avg=", "speedup_e2e(avg)=", "hip_e2e_batched(avg=", "speedup_e2e_batched(avg)=",
                    "hip_kernel(avg=", "speedup_kernel(avg)="):
            assert col in l, (col, l)


def test_run_dft_benchmark_report_detects_a_wrong_cpu_column(p3, oracle):
    def wrong(x):
        y = oracle.dft_batch(x)
        y[0, 0] ^= 1
        return y
    text = p3.run_dft_benchmark_report(cpu_dft=wrong)
    assert text == "dft benchmark failed: dft benchmark mismatch at h=256, w=8"  # fib_air.rs:193-196


def test_run_dft_benchmark_report_without_a_cpu_column(p3):
    lines = p3.run_dft_benchmark_report().splitlines()
    assert len(lines) == 12 and all("cpu(" not in l and "speedup" not in l for l in lines[1:])
