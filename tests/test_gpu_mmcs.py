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
