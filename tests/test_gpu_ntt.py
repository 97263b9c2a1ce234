"""GPU parity tests (MI355X): libp3hip NTT / coset-LDE through the C ABI vs the C oracle and the
committed golden fixtures.  Bit-exact (integer arithmetic): every comparison is array_equal."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
P = 0x78000001

# reference benchmark shapes, native/src/fib_air.rs:103-117
REF_SHAPES = [(256, 8), (1024, 8), (4096, 8), (16384, 8), (4096, 32), (16384, 32), (4096, 64), (4096, 128),
              (16384, 64), (16384, 128), (256, 16000)]


@pytest.fixture(scope="module")
def dft(p3):
    ok, msg = p3.is_available()
    assert ok, msg
    return p3.GpuDft.with_backend(p3.BackendKind.Hip)


def _rand(rng, h, w):
    return rng.integers(0, P, size=(h, w), dtype=np.uint64).astype(np.uint32)  # any word < P is a valid Monty residue


def test_golden_dft_host_path(dft, oracle):
    for case in golden("dft.json"):
        if case["input"] == "benchmark_input":
            x = oracle.benchmark_input(case["h"], case["w"])
        else:
            x = oracle.to_monty(np.array(case["input"], dtype=np.uint64))
        got = dft.dft_batch(x)
        assert np.array_equal(oracle.from_monty(got), np.array(case["out"], dtype=np.uint32)), (case["h"], case["w"])


def test_golden_coset_lde_host_path(dft, oracle):
    for case in golden("coset_lde.json"):
        x = oracle.generate_trace_rows(0, 1, case["h"]) if case["input"] == "fib_trace" else \
            oracle.benchmark_input(case["h"], case["w"])
        sh = int(oracle.to_monty(case["shift"]))
        exp = np.array(case["out"], dtype=np.uint32)
        nat = dft.coset_lde_batch(x, case["added_bits"], sh)
        assert np.array_equal(oracle.from_monty(nat), exp), case
        br = dft.coset_lde_batch(x, case["added_bits"], sh, bit_reversed_out=True)
        assert np.array_equal(oracle.bit_reverse_rows(br), nat), case


@pytest.mark.parametrize("h,w", REF_SHAPES)
def test_reference_benchmark_shapes(dft, oracle, h, w):
    # the reference's own equality check, fib_air.rs:193-196, on its benchmark_input (fib_air.rs:77-86)
    x = oracle.benchmark_input(h, w)
    assert np.array_equal(dft.dft_batch(x), oracle.dft_batch(x))


@pytest.mark.parametrize("log_h", list(range(0, 15)) + [16, 17])
@pytest.mark.parametrize("w", [1, 2, 3, 4, 5, 8, 31, 32, 33, 70])
def test_dft_all_heights_widths(dft, oracle, log_h, w):
    rng = np.random.default_rng(log_h * 100 + w)
    x = _rand(rng, 1 << log_h, w)
    exp = oracle.dft_batch(x)
    assert np.array_equal(dft.dft_batch(x), exp)
    assert np.array_equal(dft.idft_batch(exp), x)


@pytest.mark.parametrize("log_h,w,ab", [(0, 2, 1), (1, 2, 1), (3, 2, 1), (3, 2, 2), (5, 7, 3), (10, 2, 1), (11, 2, 1),
                                        (12, 4, 1), (12, 2, 2), (13, 2, 3), (16, 2, 1), (16, 4, 2), (10, 40, 1),
                                        (9, 2, 0)])
def test_coset_lde_vs_oracle(dft, oracle, p3, log_h, w, ab):
    rng = np.random.default_rng(7 * log_h + w + ab)
    x = _rand(rng, 1 << log_h, w)
    for shift in (p3.GENERATOR_MONTY, p3.MONTY_ONE, int(rng.integers(1, P))):
        exp = oracle.coset_lde_batch(x, ab, shift)
        assert np.array_equal(dft.coset_lde_batch(x, ab, shift), exp)
        assert np.array_equal(dft.coset_lde_batch(x, ab, shift, bit_reversed_out=True), oracle.bit_reverse_rows(exp))
        if ab == 0:
            coeffs = oracle.idft_batch(x)
            assert np.array_equal(dft.coset_dft_batch(coeffs, shift), oracle.coset_dft_batch(coeffs, shift))


def test_device_resident_path_and_stream(dft, oracle, p3):
    import torch
    rng = np.random.default_rng(3)
    x = _rand(rng, 1 << 14, 2)
    xd = p3.dev_u32(x)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        yd = dft.dft_batch(xd)
        ld = dft.coset_lde_batch(xd, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
        back = dft.idft_batch(yd)
    s.synchronize()
    assert np.array_equal(p3.host_u32(yd), oracle.dft_batch(x))
    assert np.array_equal(p3.host_u32(back), x)
    assert np.array_equal(p3.host_u32(ld), oracle.coset_lde_batch(x, 1, p3.GENERATOR_MONTY, True))
    assert np.array_equal(p3.host_u32(p3.bit_reverse_rows(ld)), oracle.coset_lde_batch(x, 1, p3.GENERATOR_MONTY))


def test_fib_air_headline_size_properties(dft, oracle, p3):
    """BASELINE cfg2: 2^20 x 2 fib trace, blowup 2.  Too big for the O(n log n) oracle to be quick in a
    loop, so: one full oracle comparison, plus size-independent properties."""
    import torch
    n = 1 << 20
    x = oracle.generate_trace_rows(0, 1, n)
    xd = p3.dev_u32(x)
    lde = dft.coset_lde_batch(xd, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    torch.cuda.synchronize()
    got = p3.host_u32(lde)
    assert np.array_equal(got, oracle.coset_lde_batch(x, 1, p3.GENERATOR_MONTY, True))
    # round trip: idft(dft(x)) == x on device
    assert torch.equal(dft.idft_batch(dft.dft_batch(xd)), xd)
    # linearity: dft(a) + dft(b) == dft(a + b)  (mod P, Montgomery form is linear)
    rng = np.random.default_rng(5)
    a = _rand(rng, n, 2); b = _rand(rng, n, 2)
    s = ((a.astype(np.uint64) + b) % P).astype(np.uint32)
    fa, fb, fs = (p3.host_u32(dft.dft_batch(p3.dev_u32(v))) for v in (a, b, s))
    assert np.array_equal(((fa.astype(np.uint64) + fb) % P).astype(np.uint32), fs)
    # the LDE restricted to every 2nd natural-order point of the blown-up coset... shift=1: extends x itself
    ext = p3.host_u32(dft.coset_lde_batch(xd, 1, p3.MONTY_ONE))
    assert np.array_equal(ext[::2], x)


def test_large_heights_round_trip(dft, p3):
    """2^22 (two passes of 11) and 2^24 (three passes): inverse round trip + spot-check against a direct
    evaluation of a sparse polynomial."""
    import torch
    for log_h in (22, 24):
        n = 1 << log_h
        g = torch.Generator(device="cuda").manual_seed(log_h)
        xd = torch.randint(0, P, (n, 2), dtype=torch.int32, device="cuda", generator=g)
        yd = dft.dft_batch(xd)
        assert torch.equal(dft.idft_batch(yd), xd)
        del yd
        # delta at row r: dft[k] = x_r * w^(r k) -> column is a geometric sequence; check a few entries
        r = 12345
        d = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
        d[r, 0] = p3.MONTY_ONE
        out = p3.host_u32(dft.dft_batch(d))
        w = pow(pow(31, 15, P), 1 << (27 - log_h), P)
        R = 1 << 32
        for k in (0, 1, 2, 77777, n - 1):
            assert int(out[k, 0]) == pow(w, r * k, P) * R % P
            assert int(out[k, 1]) == 0


def test_error_paths(dft, p3):
    with pytest.raises(p3.P3HipError) as e:
        dft.dft_batch(np.zeros((12, 2), np.uint32))
    assert e.value.code == -1 and "power-of-two" in e.value.message
    assert p3.take_last_error() is None  # take-and-clear
    with pytest.raises(p3.P3HipError):
        p3.GpuDft.with_backend(p3.BackendKind.Cpu).dft_batch(np.zeros((4, 2), np.uint32))
    # empty matrices are a no-op
    assert dft.dft_batch(np.zeros((0, 2), np.uint32)).shape == (0, 2)


def test_keccak_air_shaped_wide_matrix(dft, oracle, p3):
    """BASELINE configs[4] shape family: width 2633 (Keccak-f AIR trace width), non power of two, wider than a
    tile run; LDE + commitment parity on a short instance (the 2^16-row size is exercised by tools/kernel_bench)."""
    rng = np.random.default_rng(2633)
    x = _rand(rng, 1 << 7, 2633)
    exp = oracle.coset_lde_batch(x, 1, p3.GENERATOR_MONTY, True)
    got = dft.coset_lde_batch(x, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    assert np.array_equal(got, exp)
    root, _ = p3.MerkleTreeMmcs().commit([got])
    oroot, _ = oracle.mmcs_commit([exp])
    assert np.array_equal(root, oroot)


def test_randomized_shapes_sweep(dft, oracle, p3):
    """Seeded random sweep over (log height, width, blowup, shift, output order): every plan shape the
    planner can pick for heights up to 2^15 (single pass, two and three passes, fused middle, fast and
    general kernels, power-of-two and odd widths)."""
    rng = np.random.default_rng(20261004)
    for it in range(120):
        log_h = int(rng.integers(0, 16))
        w = int(rng.choice([1, 2, 3, 4, 6, 8, 16, 24, 32, 33, 48, 64, 100]))
        if (1 << log_h) * w > 1 << 20:
            w = max(1, (1 << 20) >> log_h)
        ab = int(rng.integers(0, 4))
        if log_h + ab > 17:
            ab = 17 - log_h
        shift = int(rng.choice([p3.GENERATOR_MONTY, p3.MONTY_ONE, int(rng.integers(1, P))]))
        x = _rand(rng, 1 << log_h, w)
        br = bool(rng.integers(0, 2))
        exp = oracle.coset_lde_batch(x, ab, shift, br)
        got = dft.coset_lde_batch(x, ab, shift, bit_reversed_out=br)
        assert np.array_equal(got, exp), (it, log_h, w, ab, br)
        if it % 3 == 0:
            assert np.array_equal(dft.dft_batch(x), oracle.dft_batch(x)), (it, log_h, w)


def test_randomized_large_shapes(dft, oracle, p3):
    """A handful of seeded random large cases (2^16..2^20 rows) against the oracle: three-pass plans, the fused
    middle pass (blowup 2 and 4), 9- and 10-stage tiles, natural and bit-reversed output."""
    rng = np.random.default_rng(424242)
    cases = [(16, 3, 1, True), (17, 2, 2, False), (18, 4, 1, True), (18, 1, 2, True), (19, 2, 1, False), (20, 2, 1, True),
             (20, 4, 2, True), (16, 40, 3, False), (19, 8, 0, True)]
    for log_h, w, ab, br in cases:
        x = _rand(rng, 1 << log_h, w)
        shift = int(rng.choice([p3.GENERATOR_MONTY, int(rng.integers(1, P))]))
        exp = oracle.coset_lde_batch(x, ab, shift, br)
        got = dft.coset_lde_batch(x, ab, shift, bit_reversed_out=br)
        assert np.array_equal(got, exp), (log_h, w, ab, br)
    x = _rand(rng, 1 << 19, 3)
    y = oracle.dft_batch(x)
    assert np.array_equal(dft.dft_batch(x), y)
    assert np.array_equal(dft.idft_batch(y), x)


@pytest.mark.parametrize("log_h,w,ab", [(16, 2, 1), (16, 4, 2), (16, 8, 3), (17, 2, 1), (17, 4, 1), (18, 2, 2), (18, 8, 1),
                                        (19, 2, 1), (19, 4, 3), (20, 2, 1), (20, 4, 1), (20, 2, 2), (21, 2, 1), (21, 8, 1),
                                        (22, 2, 1), (22, 4, 2), (16, 16, 1), (18, 16, 2), (19, 16, 1), (17, 2, 3), (20, 2, 3), (21, 2, 2),
                                        (23, 2, 1)])
def test_narrow_three_launch_lde(dft, oracle, p3, log_h, w, ab):
    """The two-digit, three-launch LDE for narrow matrices (ntt_narrow.hip.h; W in {2, 4, 8, 16}, 2^16..2^24 rows,
    bit-reversed output): every digit size 8..11 in both positions, every slot/row split, blowup 2, 4 and 8,
    against the oracle bit for bit.  (2^24 x 2 at blowup 4 — BASELINE configs[2]'s own shape — is the next test.)"""
    rng = np.random.default_rng(1000 * log_h + 10 * w + ab)
    x = _rand(rng, 1 << log_h, w)
    shift = p3.GENERATOR_MONTY if (log_h + w) % 2 == 0 else int(rng.integers(1, P))
    exp = oracle.coset_lde_batch(x, ab, shift, True)
    got = dft.coset_lde_batch(x, ab, shift, bit_reversed_out=True)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("log_h,ab", [(23, 1), (24, 2)])
def test_twelve_stage_digit_lde_host_and_device_paths(dft, oracle, p3, log_h, ab):
    """12-stage digits (2^23 / 2^24 rows x 2): blocked intermediates, K3 out of place with partner tiles.  (24, 2) with shift
    GENERATOR is cfg3's trace LDE itself (2^24 x 2 -> 2^26 x 2), compared element by element.  Both entry points: the
    host-pointer one stages `dst` in a context scratch buffer (c_api.hip dft_host) — K3's own scratch must not be that buffer
    (round-3 advisor finding: slot 3 was shared, K3 read what its partner tiles overwrote) — and the `_dev` one with a
    caller-owned output.  Each path twice, so the second call runs with every scratch buffer already at its final size."""
    import torch
    rng = np.random.default_rng(120000 + log_h)
    x = _rand(rng, 1 << log_h, 2)
    exp = oracle.coset_lde_batch(x, ab, p3.GENERATOR_MONTY, True)
    for rep in range(2):
        got = dft.coset_lde_batch(x, ab, p3.GENERATOR_MONTY, bit_reversed_out=True)
        assert np.array_equal(got, exp), ("host path", rep)
        del got
    dx = p3.dev_u32(x)
    for rep in range(2):
        dgot = dft.coset_lde_batch(dx, ab, p3.GENERATOR_MONTY, bit_reversed_out=True)
        torch.cuda.synchronize()
        assert np.array_equal(p3.host_u32(dgot), exp), ("device path", rep)
        del dgot


@pytest.mark.parametrize("log_h,ab", [(25, 1), (25, 2)])
def test_lde_above_the_narrow_plan_2_25_rows(dft, oracle, p3, log_h, ab):
    """2^25 rows x 2 leave the three-launch plan (ntt.hip lde_narrow: heights up to 2^24) for the general plan, at heights it had
    never run before round 5: blowup 2 (2^26 output rows) and blowup 4 (2^27 rows = the field's two-adicity, the largest LDE the
    entry point admits), element by element against the oracle (all host cores), through the device entry point."""
    import torch
    rng = np.random.default_rng(250000 + ab)
    x = _rand(rng, 1 << log_h, 2)
    oracle.set_threads(oracle.test_threads())
    try:
        exp = oracle.coset_lde_batch(x, ab, p3.GENERATOR_MONTY, True)
    finally:
        oracle.set_threads(1)
    dgot = dft.coset_lde_batch(p3.dev_u32(x), ab, p3.GENERATOR_MONTY, bit_reversed_out=True)
    torch.cuda.synchronize()
    got = p3.host_u32(dgot)
    del dgot
    assert got.shape == exp.shape
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("log_h,ab", [(16, 1), (17, 2), (18, 1), (19, 3), (20, 1), (21, 1), (22, 1)])
def test_narrow_lde_of_six_columns(dft, oracle, p3, log_h, ab):
    """W = 6 — the hiding prover's randomized trace (2 trace columns + 4 random codewords, fib_air.rs:65) — through the narrow
    plan: three column pairs (or six single columns) per row, so a tile's slots straddle rows."""
    rng = np.random.default_rng(6000 + 10 * log_h + ab)
    x = _rand(rng, 1 << log_h, 6)
    shift = p3.GENERATOR_MONTY if log_h % 2 == 0 else int(rng.integers(1, P))
    assert np.array_equal(dft.coset_lde_batch(x, ab, shift, bit_reversed_out=True), oracle.coset_lde_batch(x, ab, shift, True))


@pytest.mark.parametrize("log_h,w,ab", [(16, 4, 1), (17, 2, 2), (18, 6, 1), (19, 4, 1), (20, 4, 1), (21, 4, 1), (21, 16, 1), (22, 2, 2),
                                        (12, 4, 1), (15, 5, 2), (16, 32, 1)])
def test_coset_lde_from_coefficients(oracle, p3, log_h, w, ab):
    """p3hip_coset_lde_from_coeffs_bb31_dev: evaluations over shift*<g> of a matrix given by COEFFICIENTS, bit-reversed rows —
    two launches of the narrow plan (no inverse digits) where it covers the shape, dft + LDE elsewhere (the last three shapes).
    Oracle: evaluate on the subgroup (dft_batch), then the ordinary coset LDE."""
    rng = np.random.default_rng(7000 + 100 * log_h + 10 * w + ab)
    c = _rand(rng, 1 << log_h, w)
    shift = p3.GENERATOR_MONTY if (log_h + w) % 2 == 0 else int(rng.integers(1, P))
    exp = oracle.coset_lde_batch(oracle.dft_batch(c), ab, shift, True)
    got = p3.host_u32(p3.coset_lde_from_coeffs(p3.dev_u32(c), ab, shift))
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("log_h,w,ab", [(16, 64, 1), (16, 65, 2), (16, 100, 1), (16, 333, 1), (16, 257, 3),
                                        (17, 65, 1), (17, 100, 2), (18, 77, 1), (18, 64, 2)])
def test_wide_matrices_on_the_two_digit_plan(dft, oracle, p3, log_h, w, ab):
    """2^16..2^18 rows x >= 64 columns (any width, odd ones too): the three-launch plan with 128-byte tile rows, whose tiles straddle
    matrix rows — the shape class of BASELINE configs[4] (2^16 x 2633, tests/test_gpu_cfg5.py holds that size itself).  2^17 and
    2^18 rows run 9-stage digits on 1024-thread tiles (round 4)."""
    rng = np.random.default_rng(1000 * log_h + 10 * w + ab)
    x = _rand(rng, 1 << log_h, w)
    shift = p3.GENERATOR_MONTY if w % 2 == 0 else int(rng.integers(1, P))
    assert np.array_equal(dft.coset_lde_batch(x, ab, shift, bit_reversed_out=True), oracle.coset_lde_batch(x, ab, shift, True))


def test_raw_u32_plan_entry_like_the_reference_benchmark(p3, oracle):
    """prepare_compute_plan(width, height, 0, log_n) + setup_pipeline_plan(&plan, &[u32]) as the reference's benchmark
    drives them (fib_air.rs:128-134): natural-order Montgomery words in and out."""
    for h, w in [(256, 8), (4096, 32), (1024, 3)]:
        x = oracle.benchmark_input(h, w) if hasattr(oracle, "benchmark_input") else _rand(np.random.default_rng(h), h, w)
        plan = p3.plan.prepare_compute_plan(w, h, 0, h.bit_length() - 1)
        got = p3.plan.setup_pipeline_plan(plan, np.asarray(x, dtype=np.uint32).reshape(-1))
        assert np.array_equal(got.reshape(h, w), oracle.dft_batch(np.asarray(x, dtype=np.uint32).reshape(h, w)))
    with pytest.raises(ValueError):
        p3.plan.setup_pipeline_plan(p3.plan.prepare_compute_plan(4, 8, 0, 3), np.zeros(5, dtype=np.uint32))


def test_two_streams_one_thread_interleaved_lde(dft, oracle, p3):
    """include/p3hip.h stream contract: one host thread, two streams, interleaved `_dev` LDEs on different inputs.
    Scratch is keyed by (thread, stream), so neither transform's intermediate is touched by the other."""
    import torch
    rng = np.random.default_rng(41)
    cases = [(16, 2, 1), (17, 2, 1), (16, 4, 2), (12, 6, 1)]  # narrow plan (3 launches, scratch T) and general plans
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for log_h, w, ab in cases:
        xa, xb = _rand(rng, 1 << log_h, w), _rand(rng, 1 << log_h, w)
        da, db = p3.dev_u32(xa), p3.dev_u32(xb)
        torch.cuda.synchronize()
        outs = []
        for rep in range(4):  # alternate the two streams call by call; nothing synchronises in between
            with torch.cuda.stream(s1):
                oa = dft.coset_lde_batch(da, ab, p3.GENERATOR_MONTY, bit_reversed_out=True)
            with torch.cuda.stream(s2):
                ob = dft.coset_lde_batch(db, ab, p3.MONTY_ONE, bit_reversed_out=True)
            outs.append((oa, ob))
        torch.cuda.synchronize()
        ea = oracle.coset_lde_batch(xa, ab, p3.GENERATOR_MONTY, True)
        eb = oracle.coset_lde_batch(xb, ab, p3.MONTY_ONE, True)
        for oa, ob in outs:
            assert np.array_equal(p3.host_u32(oa), ea), (log_h, w, ab)
            assert np.array_equal(p3.host_u32(ob), eb), (log_h, w, ab)


def test_scale_table_cache_cycling(dft, oracle, p3):
    """More distinct (shift, height) pairs on one thread than the bounded table cache holds (256 entries), then a
    blowup-8 narrow LDE (8 scale tables fetched in one call) against the oracle: the cache is only ever emptied at
    the top of a call, never between two fetches."""
    import torch
    rng = np.random.default_rng(43)
    x16 = p3.dev_u32(_rand(rng, 1 << 16, 2))
    x12 = p3.dev_u32(_rand(rng, 1 << 12, 2))
    for k in range(300):
        shift = int(oracle.to_monty(np.array([3 + k], dtype=np.uint64))[0])
        dft.coset_lde_batch(x16 if k % 2 else x12, 1, shift, bit_reversed_out=True)
    xh = _rand(rng, 1 << 16, 2)
    shift = int(rng.integers(1, P))
    got = dft.coset_lde_batch(p3.dev_u32(xh), 3, shift, bit_reversed_out=True)
    torch.cuda.synchronize()
    assert np.array_equal(p3.host_u32(got), oracle.coset_lde_batch(xh, 3, shift, True))
    # and once more straight after the cache has been refilled past its cap again
    for k in range(260):
        dft.coset_lde_batch(x12, 1, int(oracle.to_monty(np.array([1000 + k], dtype=np.uint64))[0]), bit_reversed_out=True)
    got = dft.coset_lde_batch(p3.dev_u32(xh), 3, shift, bit_reversed_out=True)
    assert np.array_equal(p3.host_u32(got), oracle.coset_lde_batch(xh, 3, shift, True))


def test_release_thread_context_and_reuse(dft, oracle, p3):
    """p3hip_release_thread_context frees the thread's tables and scratch; the next call rebuilds them."""
    from plonky3_mobile_amd import _lib
    rng = np.random.default_rng(47)
    x = _rand(rng, 1 << 10, 3)
    exp = oracle.dft_batch(x)
    assert np.array_equal(dft.dft_batch(x), exp)
    _lib.lib().p3hip_release_thread_context()
    assert np.array_equal(dft.dft_batch(x), exp)
    assert np.array_equal(p3.host_u32(dft.dft_batch(p3.dev_u32(x))), exp)


def test_per_call_timing_line(dft, oracle, p3):
    """backend_vulkan.rs:1385-1423: one line per DFT call with upload / stages / readback / total and the GPU timestamps."""
    import re
    x = oracle.benchmark_input(4096, 8)
    dft.dft_batch(x)
    line = p3.last_timing_line()
    m = re.match(r"hip dft: op=dft h=4096 w=8 stages=12 upload=([0-9.]+)ms stages=([0-9.]+)ms readback=([0-9.]+)ms total=([0-9.]+)ms "
                 r"gpu\(stage=([0-9.]+)ms copy_back=([0-9.]+)ms total=([0-9.]+)ms\)$", line)
    assert m, line
    up, st, rb, tot, gs, gc, gt = map(float, m.groups())
    assert tot >= rb and gt >= gs and gs > 0
    dft.coset_lde_batch(x, 1, p3.GENERATOR_MONTY)
    assert p3.last_timing_line().startswith("hip dft: op=coset_lde h=4096 w=8 stages=13 ")
