"""CPU tests: the C oracle (oracle/) against the committed golden fixtures (tests/golden/,
big-int mathematics + Plonky3's own Poseidon2 known-answer vector)."""
import numpy as np
import pytest

from conftest import golden

P = 0x78000001


def M(o, x):
    return o.to_monty(np.asarray(x, dtype=np.uint64))


def test_field_ops(oracle):
    g = golden("field_ops.json")
    L = oracle.lib()
    assert g["p"] == P
    assert L.p3o_to_monty(1) == g["monty_r_mod_p"] == 0x0FFFFFFE
    assert L.p3o_mul(L.p3o_to_monty(1), L.p3o_to_monty(1)) == L.p3o_to_monty(1)
    for c in g["cases"]:
        a, b = L.p3o_to_monty(c["a"]), L.p3o_to_monty(c["b"])
        assert a == c["monty_a"]
        assert L.p3o_from_monty(L.p3o_add(a, b)) == c["add"]
        assert L.p3o_from_monty(L.p3o_sub(a, b)) == c["sub"]
        assert L.p3o_from_monty(L.p3o_mul(a, b)) == c["mul"]
        if c["a"]:
            assert L.p3o_from_monty(L.p3o_inv(a)) == c["inv_a"]
    for bits, v in g["two_adic_generator"].items():
        assert L.p3o_from_monty(L.p3o_two_adic_generator(int(bits))) == v
    # two_adic_generator(27) = 31^15
    assert g["two_adic_generator"]["27"] == 0x1A427A41


def test_ext_field(oracle):
    rng = np.random.default_rng(1)
    L = oracle.lib()
    for _ in range(20):
        a = M(oracle, rng.integers(0, P, 4))
        out = np.zeros(4, np.uint32)
        inv = np.zeros(4, np.uint32)
        L.p3o_ext_inv(oracle._p(a), oracle._p(inv))
        L.p3o_ext_mul(oracle._p(a), oracle._p(inv), oracle._p(out))
        assert list(oracle.from_monty(out)) == [1, 0, 0, 0]
    # x * x^3 = x^4 = 11
    x = M(oracle, [0, 1, 0, 0]); x3 = M(oracle, [0, 0, 0, 1])
    out = np.zeros(4, np.uint32)
    L.p3o_ext_mul(oracle._p(x), oracle._p(x3), oracle._p(out))
    assert list(oracle.from_monty(out)) == [11, 0, 0, 0]


def _input(o, case):
    if case["input"] == "benchmark_input":
        return o.benchmark_input(case["h"], case["w"])
    if case["input"] == "fib_trace":
        return o.generate_trace_rows(0, 1, case["h"])
    return M(o, case["input"])


def test_dft_golden(oracle):
    for case in golden("dft.json"):
        x = _input(oracle, case)
        exp = np.array(case["out"], dtype=np.uint32)
        assert np.array_equal(oracle.from_monty(oracle.dft_batch(x)), exp), (case["h"], case["w"])
        assert np.array_equal(oracle.from_monty(oracle.naive_dft(x)), exp)


def test_dft_reference_shapes_roundtrip(oracle):
    # reference shapes fib_air.rs:103-117 (the small ones); idft(dft(x)) == x
    for h, w in [(256, 8), (1024, 8), (4096, 8), (4096, 32), (256, 1000)]:
        x = oracle.benchmark_input(h, w)
        assert np.array_equal(oracle.idft_batch(oracle.dft_batch(x)), x)


def test_dft_rejects_non_pow2(oracle):
    with pytest.raises(ValueError):
        oracle.dft_batch(np.zeros((12, 2), np.uint32))


def test_twiddle_table_layout(oracle):
    # backend_vulkan.rs:977-996: stage s at offset 2^s-1, entries step^i
    log_n = 6
    tw = oracle.from_monty(oracle.twiddle_table(log_n))
    g = golden("field_ops.json")["two_adic_generator"]
    for s in range(log_n):
        step = pow(g[str(log_n)], 1 << (log_n - s - 1), P)
        assert [int(v) for v in tw[(1 << s) - 1:(1 << (s + 1)) - 1]] == [pow(step, i, P) for i in range(1 << s)]


def test_coset_lde_golden(oracle):
    for case in golden("coset_lde.json"):
        x = _input(oracle, case)
        exp = np.array(case["out"], dtype=np.uint32)
        sh = oracle.lib().p3o_to_monty(case["shift"])
        nat = oracle.coset_lde_batch(x, case["added_bits"], sh)
        assert np.array_equal(oracle.from_monty(nat), exp), case["h"]
        br = oracle.coset_lde_batch(x, case["added_bits"], sh, bit_reversed_out=True)
        assert np.array_equal(oracle.bit_reverse_rows(br), nat)


def test_fib_trace(oracle):
    g = golden("fib_trace.json")
    t = oracle.from_monty(oracle.generate_trace_rows(g["a"], g["b"], g["n"]))
    assert t.tolist() == g["rows"]
    assert g["n8_last_right"] == 21  # fib_air.rs:56-57: n = 8, x = 21


def test_poseidon2_upstream_kat(oracle):
    g = golden("poseidon2_bb16_kat.json")
    rc = (M(oracle, g["rc_ext_init"]), M(oracle, g["rc_internal"]), M(oracle, g["rc_ext_final"]))
    out = oracle.poseidon2_permute(M(oracle, g["input"]), rc)
    assert oracle.from_monty(out).tolist() == g["expected"]


def test_poseidon2_default_constants(oracle):
    g = golden("poseidon2_bb16_default.json")
    assert g["rc_ext_init_row0"][0] == 0x69CBB6AF and g["rc_internal"][0] == 0x5A8053C0
    for c in g["cases"]:
        out = oracle.poseidon2_permute(M(oracle, c["input"]))
        assert oracle.from_monty(out).tolist() == c["expected"]


def test_sponge_compress_merkle(oracle):
    g = golden("mmcs.json")
    for c in g["hash_row"]:
        assert oracle.from_monty(oracle.hash_row(M(oracle, c["items"]))).tolist() == c["digest"]
    assert g["hash_row"][0]["digest"] == [0] * 8  # empty input: no permutation
    for c in g["compress"]:
        assert oracle.from_monty(oracle.compress(M(oracle, c["l"]), M(oracle, c["r"]))).tolist() == c["digest"]
    for t in g["trees"]:
        mats = [M(oracle, m).reshape(h, w) for m, (h, w) in zip(t["mats"], t["dims"])]
        root, tree = oracle.mmcs_commit(mats)
        layers = tree.layers()
        assert len(layers) == len(t["layers"])
        for got, exp in zip(layers, t["layers"]):
            assert oracle.from_monty(got).tolist() == exp
        assert oracle.from_monty(root).tolist() == t["layers"][-1][0]
        maxh = max(h for h, _ in t["dims"])
        for idx in range(maxh):
            rows, path = tree.open_batch(idx)
            assert oracle.mmcs_verify_batch(root, t["dims"], idx, rows, path)
            if rows.size:
                bad = rows.copy(); bad[0] ^= 1
                assert not oracle.mmcs_verify_batch(root, t["dims"], idx, bad, path)
