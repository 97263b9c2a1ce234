"""GPU parity tests: Poseidon2 permutation and the Merkle-tree MMCS vs oracle + golden fixtures."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu
P = 0x78000001


def _rand(rng, h, w):
    return rng.integers(0, P, size=(h, w), dtype=np.uint64).astype(np.uint32)


def test_poseidon2_default_constants_golden(p3, oracle):
    g = golden("poseidon2_bb16_default.json")
    ins = oracle.to_monty(np.array([c["input"] for c in g["cases"]], dtype=np.uint64))
    out = p3.poseidon2_permute(ins)
    assert oracle.from_monty(out).tolist() == [c["expected"] for c in g["cases"]]


def test_poseidon2_vs_oracle_random(p3, oracle):
    rng = np.random.default_rng(11)
    st = _rand(rng, 5000, 16)
    # edge values: 0, P-1 and low-bit patterns exercise the 2^-k halving tricks
    st[0] = 0; st[1] = P - 1; st[2] = np.arange(16); st[3] = (1 << np.arange(16)).astype(np.uint32)
    st[4] = P - 1 - np.arange(16)
    got = p3.poseidon2_permute(st)
    exp = np.stack([oracle.poseidon2_permute(s) for s in st])
    assert np.array_equal(got, exp)
    d = p3.dev_u32(st)
    p3.poseidon2_permute(d)
    assert np.array_equal(p3.host_u32(d), exp)


def test_mmcs_golden_trees(p3, oracle):
    mm = p3.MerkleTreeMmcs()
    for t in golden("mmcs.json")["trees"]:
        mats = [oracle.to_monty(np.array(m, dtype=np.uint64)).reshape(h, w) for m, (h, w) in zip(t["mats"], t["dims"])]
        root, tree = mm.commit(mats)
        layers = tree.digest_layers()
        assert [oracle.from_monty(l).tolist() for l in layers] == t["layers"]
        assert oracle.from_monty(root).tolist() == t["layers"][-1][0]


@pytest.mark.parametrize("dims", [[(1, 2)], [(2, 2)], [(1 << 10, 2)], [(1 << 12, 4)], [(1 << 9, 8), (1 << 9, 3)],
                                  [(1 << 11, 2), (1 << 10, 4), (1 << 3, 9)], [(1 << 8, 8)], [(1 << 6, 100)],
                                  [(1 << 13, 17), (1 << 12, 1)], [(512, 2), (256, 2), (128, 2), (1, 2)]])
def test_mmcs_commit_open_vs_oracle(p3, oracle, dims):
    rng = np.random.default_rng(sum(h * w for h, w in dims))
    mats = [_rand(rng, h, w) for h, w in dims]
    mm = p3.MerkleTreeMmcs()
    root, tree = mm.commit(mats)
    oroot, otree = oracle.mmcs_commit(mats)
    assert np.array_equal(root, oroot)
    for a, b in zip(tree.digest_layers(), otree.layers()):
        assert np.array_equal(a, b)
    maxh = max(h for h, _ in dims)
    for idx in sorted({0, 1 % maxh, maxh // 2, maxh - 1, int(rng.integers(0, maxh))}):
        rows, path = mm.open_batch(idx, tree)
        orows, opath = otree.open_batch(idx)
        assert np.array_equal(np.concatenate(rows), orows)
        assert np.array_equal(path, opath)
        assert oracle.mmcs_verify_batch(root, dims, idx, np.concatenate(rows), path)


def test_mmcs_headline_size(p3, oracle):
    """BASELINE cfg2 trace commitment: bit-reversed LDE 2^21 x 2 committed on device; root checked against
    the oracle, openings checked with the oracle's verify_batch (size-independent property)."""
    x = oracle.generate_trace_rows(0, 1, 1 << 20)
    dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
    lde = dft.coset_lde_batch(p3.dev_u32(x), 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    mm = p3.MerkleTreeMmcs()
    root, tree = mm.commit([lde])
    host = p3.host_u32(lde)
    oroot, _ = oracle.mmcs_commit([host])
    assert np.array_equal(root, oroot)
    for idx in (0, 1, 12345, (1 << 21) - 1):
        rows, path = mm.open_batch(idx, tree)
        assert np.array_equal(rows[0], host[idx])
        assert oracle.mmcs_verify_batch(root, [(1 << 21, 2)], idx, rows[0], path)


@pytest.mark.parametrize("hash", ["poseidon2", "keccak"])
@pytest.mark.parametrize("h,w", [(4096, 64), (4096, 77), (8192, 333), (16384, 65), (4096, 2633), (4096, 68), (4096, 102), (4096, 103)])
def test_wide_rows_leaf_layer_vs_oracle(p3, oracle, h, w, hash):
    """Wide rows take the LDS-staged leaf kernels (leaf_hash_f64_wide_kernel from 64 words: 32-word chunks per row;
    keccak_leaf_wide_kernel from 68 words: one 34-word rate block per step; partial last chunks, odd widths, widths that are exact
    multiples of the block): every leaf digest and the root against the oracle's sponge."""
    rng = np.random.default_rng(100 * w + h)
    m = _rand(rng, h, w)
    kind = oracle.HASH_KECCAK if hash == "keccak" else oracle.HASH_POSEIDON2
    root, tree = p3.MerkleTreeMmcs(hash).commit([m])
    leaves = tree.digest_layers()[0]
    rows = np.concatenate([np.arange(0, 300), rng.integers(0, h, 200), np.arange(h - 260, h)])
    hr = oracle.keccak_hash_row if hash == "keccak" else oracle.hash_row
    exp = np.stack([hr(m[r]) for r in rows])
    assert np.array_equal(leaves[rows], exp)
    oroot, _ = oracle.mmcs_commit([m], kind)
    assert np.array_equal(root, oroot)
    tree.free()


def test_mmcs_rejects_bad_input(p3):
    mm = p3.MerkleTreeMmcs()
    with pytest.raises(p3.P3HipError):
        mm.commit([np.zeros((12, 2), np.uint32)])


def test_mmcs_randomized_dims(p3, oracle):
    """Seeded random commitments: 1-4 matrices of random power-of-two heights and random widths (injection
    at arbitrary layers, lane-cooperative and one-state-per-lane layers mixed), roots and openings vs oracle."""
    rng = np.random.default_rng(77)
    mm = p3.MerkleTreeMmcs()
    for it in range(40):
        k = int(rng.integers(1, 5))
        dims = [(1 << int(rng.integers(0, 17)), int(rng.integers(1, 20))) for _ in range(k)]
        dims = [(h, min(w, max(1, (1 << 18) // h))) for h, w in dims]
        mats = [_rand(rng, h, w) for h, w in dims]
        root, tree = mm.commit(mats)
        oroot, otree = oracle.mmcs_commit(mats)
        assert np.array_equal(root, oroot), (it, dims)
        maxh = max(h for h, _ in dims)
        idx = int(rng.integers(0, maxh))
        rows, path = mm.open_batch(idx, tree)
        orows, opath = otree.open_batch(idx)
        assert np.array_equal(np.concatenate(rows), orows) and np.array_equal(path, opath), (it, dims, idx)
        tree.free()


def _probe(p3, states_f64, mode):
    """p3hip_poseidon2_f64_probe_dev: integer-valued doubles in, canonical values out (numpy uint32, canonical form)."""
    import ctypes as C
    import torch
    from plonky3_mobile_amd import _lib
    d = torch.from_numpy(np.ascontiguousarray(states_f64, dtype=np.float64)).cuda()
    out = torch.empty(d.shape, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib().p3hip_poseidon2_f64_probe_dev(C.c_void_p(d.data_ptr()), C.c_void_p(out.data_ptr()), d.shape[0], mode,
                                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint32)


def test_fp64_reduce_at_rounding_ties(p3, oracle):
    """reduce(x) = x - rint(x / P) P for integers x straddling the half-way points ((2k+1) P +- 1) / 2 (where the
    quotient estimate's fraction is closest to .5), at every magnitude up to 2^52, and at +-(2^52 .. 2^53): the result
    must be congruent to x, whatever way the tie falls."""
    xs = []
    for k in list(range(0, 40)) + [2 ** e for e in range(6, 22)] + [2 ** 21 + 12345, 2 ** 21 - 1]:
        for d in (-3, -1, 1, 3):
            x = ((2 * k + 1) * P + d) // 2
            if x < 2 ** 52:
                xs += [x, -x]
    xs += [2 ** 52 - 1, -(2 ** 52 - 1), 2 ** 52 + 2, 15 ** 4 * 35 * (P // 2 + 1), -(15 ** 4 * 35 * (P // 2 + 1))]
    xs += [0, 1, -1, P, -P, P - 1, P // 2, P // 2 + 1, -(P // 2), -(P // 2 + 1)]
    while len(xs) % 16:
        xs.append(0)
    arr = np.array(xs, dtype=np.float64).reshape(-1, 16)
    assert all(int(v) == x for v, x in zip(arr.reshape(-1), xs))  # all exactly representable
    got = oracle.from_monty(_probe(p3, arr, 2)).reshape(-1)
    assert got.tolist() == [x % P for x in xs]


def test_fp64_internal_rounds_at_the_largest_magnitudes(p3, oracle):
    """The 13 internal rounds of the fp64 form on states as large as the external layer can hand them (|v| up to
    35 (P/2 + 1) < 2^36, a bound no u32 input reaches deterministically), with the sign patterns that maximise the
    integer-multiplier lanes 1, 2, 4, 5, 7, 8, 12, 15 (they grow by up to 15x per round between two folds) and the
    lane sum, against exact big-integer arithmetic."""
    import pyref
    M = 35 * (P // 2 + 1)
    diag_sign = [-1, 1, 1, 1, 1, 1, -1, -1, -1, 1, 1, 1, -1, -1, -1, 1]  # sign of V[i]: aligned signs add up
    rng = np.random.default_rng(5)
    states = [[M] * 16, [-M] * 16, [M * s for s in diag_sign], [-M * s for s in diag_sign],
              [M if i in (1, 2, 4, 5, 7, 8, 12, 15) else 0 for i in range(16)],
              [(M if i % 2 else -M) for i in range(16)], [M - i for i in range(16)], [-(M - 7 * i) for i in range(16)]]
    for _ in range(200):
        mag = rng.integers(M - 2 ** 20, M + 1, size=16)
        states.append([int(m) * int(s) for m, s in zip(mag, rng.choice([-1, 1], size=16))])
    for _ in range(200):
        states.append([int(v) for v in rng.integers(-M, M + 1, size=16)])
    arr = np.array(states, dtype=np.float64)
    got = oracle.from_monty(_probe(p3, arr, 1))
    _, it, _ = pyref.DEFAULT_RC
    for row, st in zip(got, states):
        s = [v % P for v in st]
        for r in range(13):
            s[0] = pow((s[0] + it[r]) % P, 7, P)
            s = pyref._int(s)
        assert row.tolist() == s


def test_fp64_and_int32_forms_agree_on_a_million_states(p3, oracle):
    """The two arithmetic forms of the permutation (variant 0: int32 Montgomery, variant 1: exact integers in fp64) word
    for word on 2^20 random states plus structured extremes; a sample against the oracle."""
    import ctypes as C
    import torch
    from plonky3_mobile_amd import _lib
    rng = np.random.default_rng(21)
    st = _rand(rng, 1 << 20, 16)
    st[0] = 0; st[1] = P - 1; st[2] = P // 2; st[3] = P // 2 + 1
    st[4] = [(P - 1) if i in (1, 2, 4, 5, 7, 8, 12, 15) else 0 for i in range(16)]
    st[5] = [(P - 1) if i % 2 else 1 for i in range(16)]
    a, b = p3.dev_u32(st), p3.dev_u32(st)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(_lib.lib().p3hip_poseidon2_permute_variant_dev(C.c_void_p(a.data_ptr()), st.shape[0], 0, sp))
    _lib.check(_lib.lib().p3hip_poseidon2_permute_variant_dev(C.c_void_p(b.data_ptr()), st.shape[0], 1, sp))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    got = p3.host_u32(b[:64])
    assert np.array_equal(got, np.stack([oracle.poseidon2_permute(s) for s in st[:64]]))
    # the same permutation through the probe on canonical doubles (mode 0)
    canon = oracle.from_monty(st[:256]).astype(np.float64)
    assert np.array_equal(_probe(p3, canon, 0), p3.host_u32(b[:256]))
