"""Self-check run by tests/test_gpu_env_variants.py in a child process whose environment selects a non-default kernel
path (the switches are read once per process): Merkle trees of both hash configurations layer by layer against the
oracle, over heights that cross every cooperative / per-lane threshold, and one proof per configuration byte for byte."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402
from oracle import oracle as o  # noqa: E402

p3 = load_package()
o.build()
ok, msg = p3.is_available()
assert ok, msg
P = 0x78000001
rng = np.random.default_rng(7)
for hash_name, kind in (("poseidon2", o.HASH_POSEIDON2), ("keccak", o.HASH_KECCAK)):
    for dims in ([(1 << 16, 2)], [(1 << 13, 3), (1 << 9, 5), (8, 2)], [(64, 9)]):
        mats = [rng.integers(0, P, d, dtype=np.uint32) for d in dims]
        mm = p3.MerkleTreeMmcs(hash=hash_name)
        root, tree = mm.commit(mats)
        oroot, otree = o.mmcs_commit(mats, kind)
        assert np.array_equal(root, oroot), (hash_name, dims)
        for gl, ol in zip(tree.digest_layers(), otree.layers()):
            assert np.array_equal(gl, ol), (hash_name, dims, len(gl))
        tree.free()
    gfp, ofp = p3.FriParameters(1, 0, 10, 6), o.FriParams(1, 0, 10, 6)
    pr = p3.FibAirProver(13, params=gfp, hash=hash_name)
    assert pr.prove(2, 3) == o.prove_fib_air(2, 3, 13, ofp, hash=kind), hash_name
    pr.close()
hp = p3.FibAirProver(9, params=p3.FriParameters(1, 0, 8, 4), hash="keccak", hiding=True, seed=1)
assert hp.prove(0, 1) == o.prove_fib_air_hiding(0, 1, 9, o.FriParams(1, 0, 8, 4), hash=o.HASH_KECCAK, seed=1)
hp.close()
print("variant ok")
