"""Seeded random SHAPES at the two boundary calls the reference makes — `coset_lde_batch` / `dft_batch` (native/src/gpu_dft.rs:97-114) and the MMCS
commit / open (native/src/fib_air.rs:28-45) — beyond the fixed lists of test_gpu_ntt.py / test_gpu_mmcs.py: heights 2^0..2^15 with any width
1..48 (every plan of the planner, odd widths, widths that are not a whole number of 128-byte lines), 0..3 added bits, three kinds of shift, both
output orders; commitments of 1-4 matrices of unrelated heights under BOTH hashes and BOTH thread profiles.  The same cases every run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0x78000001


def _rand(rng, h, w):
    return rng.integers(0, P, size=(h, w), dtype=np.uint64).astype(np.uint32)  # any word < P is a valid Montgomery residue


def test_random_transform_shapes_equal_the_oracle(p3, oracle):
    dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
    rng = np.random.default_rng(4242)
    for it in range(90):
        log_h = int(rng.integers(0, 16))
        w = int(rng.integers(1, 49))
        w = min(w, max(1, (1 << 19) >> log_h))
        ab = int(rng.integers(0, 4))
        x = _rand(rng, 1 << log_h, w)
        shift = (p3.GENERATOR_MONTY, p3.MONTY_ONE, int(rng.integers(1, P)))[int(rng.integers(0, 3))]
        exp = oracle.coset_lde_batch(x, ab, shift)
        assert np.array_equal(dft.coset_lde_batch(x, ab, shift), exp), (it, log_h, w, ab, shift)
        assert np.array_equal(dft.coset_lde_batch(x, ab, shift, bit_reversed_out=True), oracle.bit_reverse_rows(exp)), (it, log_h, w, ab, shift, "bit-reversed")
        if it % 3 == 0:
            y = oracle.dft_batch(x)
            assert np.array_equal(dft.dft_batch(x), y), (it, log_h, w, "dft")
            assert np.array_equal(dft.idft_batch(y), x), (it, log_h, w, "idft")
            assert np.array_equal(dft.coset_dft_batch(x, shift), oracle.coset_dft_batch(x, shift)), (it, log_h, w, "coset_dft")


@pytest.mark.parametrize("hash_name", ["poseidon2", "keccak"])
@pytest.mark.parametrize("profile", ["latency", "throughput"])
def test_random_commitments_equal_the_oracle(p3, oracle, hash_name, profile):
    kind = oracle.HASH_KECCAK if hash_name == "keccak" else oracle.HASH_POSEIDON2
    rng = np.random.default_rng(99 if hash_name == "keccak" else 98)
    before = p3.get_thread_profile()
    p3.set_thread_profile(profile)
    try:
        mm = p3.MerkleTreeMmcs(hash_name)
        for it in range(30):
            k = int(rng.integers(1, 5))
            dims = [(1 << int(rng.integers(0, 17)), int(rng.integers(1, 24))) for _ in range(k)]
            dims = [(h, min(w, max(1, (1 << 18) // h))) for h, w in dims]
            mats = [_rand(rng, h, w) for h, w in dims]
            root, tree = mm.commit(mats)
            oroot, otree = oracle.mmcs_commit(mats, kind)
            assert np.array_equal(root, oroot), (it, dims)
            maxh = max(h for h, _ in dims)
            for idx in (0, maxh - 1, int(rng.integers(0, maxh))):
                rows, path = mm.open_batch(idx, tree)
                orows, opath = otree.open_batch(idx)
                assert np.array_equal(np.concatenate(rows), orows) and np.array_equal(path, opath), (it, dims, idx)
                assert oracle.mmcs_verify_batch(root, dims, idx, np.concatenate(rows), path, kind=kind), (it, dims, idx)
            tree.free()
    finally:
        p3.set_thread_profile(before)
