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
if os.environ.get("P3HIP_FRI_TAIL") == "1":
    # the single-launch FRI tail: proofs that are ALL tail (LDE of at most 2^8 rows), proofs with a final polynomial, blowup 4 and 8
    for log_n, t in ((1, (1, 0, 4, 2)), (3, (2, 2, 6, 5)), (6, (1, 0, 9, 5)), (7, (1, 3, 9, 0)), (9, (3, 1, 4, 10)), (10, (1, 0, 20, 8)), (12, (2, 0, 10, 4))):
        pr = p3.FibAirProver(log_n, params=p3.FriParameters(*t))
        assert pr.prove(3, 5) == o.prove_fib_air(3, 5, log_n, o.FriParams(*t)), ("fri tail", log_n, t)
        pr.close()
# the narrow three-launch coset LDE (2^16 rows and up) in whichever arithmetic the environment selects: integer or fp64
# butterflies (P3HIP_NTT_NARROW_F64), column pairs or single columns, one or two LDS tiles
dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
shapes = [(16, 2, 1), (17, 4, 2), (18, 8, 1), (19, 2, 3), (20, 2, 1), (21, 2, 1), (22, 4, 1), (16, 6, 1), (16, 77, 1)]
if os.environ.get("P3HIP_VARIANT_BIG_LDE") == "1":
    shapes.append((23, 2, 1))  # 12-stage digits
if os.environ.get("P3HIP_VARIANT_CFG3_LDE") == "1":
    shapes.append((24, 2, 2))  # BASELINE configs[2]'s trace LDE, element-wise, under this variant's switches
for log_h, w, ab in shapes:
    x = rng.integers(0, P, (1 << log_h, w), dtype=np.uint32)
    got = p3.host_u32(dft.coset_lde_batch(p3.dev_u32(x), ab, p3.GENERATOR_MONTY, bit_reversed_out=True))
    assert np.array_equal(got, o.coset_lde_batch(x, ab, p3.GENERATOR_MONTY, True)), ("lde", log_h, w, ab)
    del got
hp = p3.FibAirProver(9, params=p3.FriParameters(1, 0, 8, 4), hash="keccak", hiding=True, seed=1)
assert hp.prove(0, 1) == o.prove_fib_air_hiding(0, 1, 9, o.FriParams(1, 0, 8, 4), hash=o.HASH_KECCAK, seed=1)
hp.close()
# the device SmallRng stream in whichever form the environment selects (chunks per lane, small-fill kernel or not), across sizes that
# leave the last wave of a fill partly filled
dr, host = p3.DeviceRng(1), o.rng_seed_from_u64(1)
for n in (5, 700, 6371, 6372, 40000, 300001):
    got = p3.host_u32(dr.fill_field(n))
    assert np.array_equal(got, o.rng_fill_field(host, n)), ("rng", n)
    assert dr.state() == list(host), ("rng state", n)
dr.close()
print("variant ok")
