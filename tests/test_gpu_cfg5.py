"""BASELINE configs[4] at its size: the wide trace 2^16 x 2633 (the Keccak-f AIR shape; the AIR itself is not in the
reference, so the matrix is the reference's `benchmark_input`, native/src/fib_air.rs:77-86), blowup 2, bit-reversed
coset LDE + Poseidon2 MMCS commit of the 2^17-row result.

Until round 5 the full-size O(h w log h) oracle transform took minutes on one thread, so parity at this size rested on what the
domain offers (the tests below); with the oracle's transforms threaded over butterflies (oracle/dft.c) the LAST test compares the
complete LDE and both commitments with the oracle on every host core.  The properties: columns are independent (the LDE of an extracted column subset through the oracle must equal those columns of
the wide result), the shift-1 LDE reproduces its input on the even rows, idft(dft) is the identity, and openings of
the committed tree verify against the root with the ORACLE's verify_batch (which hashes the 2633-word row itself)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LOG_H, W = 16, 2633
COLS = [0, 1, 1316, 2631, 2632]


@pytest.fixture(scope="module")
def wide(p3):
    import torch
    ok, msg = p3.is_available()
    assert ok, msg
    x = p3.benchmark_input(1 << LOG_H, W)
    xd = p3.dev_u32(x)
    torch.cuda.synchronize()
    return x, xd


def test_cfg5_lde_column_subset_equals_oracle(wide, oracle, p3):
    import torch
    x, xd = wide
    dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
    lde = dft.coset_lde_batch(xd, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    torch.cuda.synchronize()
    assert tuple(lde.shape) == (2 << LOG_H, W)
    got = p3.host_u32(lde[:, COLS].contiguous())
    exp = oracle.coset_lde_batch(np.ascontiguousarray(x[:, COLS]), 1, p3.GENERATOR_MONTY, True)
    assert np.array_equal(got, exp)
    # natural order of the same transform: the bit-reversal of the committed order
    nat = dft.coset_lde_batch(xd, 1, p3.GENERATOR_MONTY)
    assert np.array_equal(p3.host_u32(nat[:, COLS].contiguous()), oracle.bit_reverse_rows(exp))
    del lde, nat
    torch.cuda.empty_cache()


def test_cfg5_shift_one_even_rows_and_round_trip(wide, oracle, p3):
    import torch
    x, xd = wide
    dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
    # LDE over the subgroup itself (shift 1): the even rows of the natural-order result are the input rows
    nat = dft.coset_lde_batch(xd, 1, p3.MONTY_ONE)
    assert torch.equal(nat[0::2], xd)
    # ... and in committed (bit-reversed) order they are the first half, the input in bit-reversed row order
    br = dft.coset_lde_batch(xd, 1, p3.MONTY_ONE, bit_reversed_out=True)
    assert torch.equal(br[: 1 << LOG_H], p3.bit_reverse_rows(xd))
    del nat, br
    y = dft.dft_batch(xd)
    assert np.array_equal(p3.host_u32(y[:, COLS].contiguous()), oracle.dft_batch(np.ascontiguousarray(x[:, COLS])))
    back = dft.idft_batch(y)
    assert torch.equal(back, xd)
    del y, back
    torch.cuda.empty_cache()


@pytest.mark.parametrize("hash", ["poseidon2", "keccak"])
def test_cfg5_commit_openings_verify_with_oracle(wide, oracle, p3, hash):
    import torch
    x, xd = wide
    dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
    lde = dft.coset_lde_batch(xd, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    mmcs = p3.MerkleTreeMmcs(hash)
    kind = oracle.HASH_KECCAK if hash == "keccak" else oracle.HASH_POSEIDON2
    root, tree = mmcs.commit([lde])
    H = 2 << LOG_H
    dims = [(H, W)]
    for index in (0, 1, 77777, H // 2, H - 1):
        rows, path = mmcs.open_batch(index, tree)
        assert np.array_equal(rows[0], p3.host_u32(lde[index]))
        assert oracle.mmcs_verify_batch(root, dims, index, rows[0], path, kind=kind)
        bad = rows[0].copy()
        bad[1316] ^= 1
        assert not oracle.mmcs_verify_batch(root, dims, index, bad, path, kind=kind)
    # the leaf layer itself, for a few rows, against the oracle's row hash
    leaves = tree.digest_layers()[0]
    for index in (0, 12345, H - 1):
        row = p3.host_u32(lde[index])
        exp = oracle.keccak_hash_row(row) if hash == "keccak" else oracle.hash_row(row)
        assert np.array_equal(leaves[index], exp)
    tree.free()
    del lde
    torch.cuda.empty_cache()


def test_cfg5_full_size_lde_and_roots_equal_oracle(wide, oracle, p3):
    """BASELINE configs[4] at its own size, element for element: the bit-reversed coset LDE of ALL 2633 columns (2^17 x 2633 words)
    equals the oracle's, and the Poseidon2 and Keccak commitments of that matrix (43.3 M leaf permutations + 131 071 compressions each)
    equal the oracle's roots.  The oracle runs on every host core (seconds per piece on the GPU box; ~4 GB of host memory)."""
    import torch
    x, xd = wide
    dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
    lde = dft.coset_lde_batch(xd, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    torch.cuda.synchronize()
    got = p3.host_u32(lde)
    oracle.set_threads(oracle.test_threads())
    try:
        exp = oracle.coset_lde_batch(x, 1, p3.GENERATOR_MONTY, True)
        assert got.shape == exp.shape
        if not np.array_equal(got, exp):
            bad = np.argwhere(got != exp)
            pytest.fail("cfg5 LDE differs from the oracle at %d of %d words, first at (row %d, column %d)" % (len(bad), got.size, bad[0][0], bad[0][1]))
        del got
        for hash_name, kind in (("poseidon2", oracle.HASH_POSEIDON2), ("keccak", oracle.HASH_KECCAK)):
            root, tree = p3.MerkleTreeMmcs(hash_name).commit([lde])
            oroot, otree = oracle.mmcs_commit([exp], kind)
            assert np.array_equal(root, oroot), hash_name
            top = tree.digest_layers()
            for gl, ol in zip(top[-8:], otree.layers()[-8:]):  # the last eight layers as well: 128 .. 1 digests
                assert np.array_equal(gl, ol), (hash_name, len(gl))
            tree.free()
            del otree
    finally:
        oracle.set_threads(1)
    del lde, exp
    torch.cuda.empty_cache()
