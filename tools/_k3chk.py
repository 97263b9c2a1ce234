import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import load_package
from oracle import oracle as o
p3 = load_package(); o.build()
dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
rng = np.random.default_rng(5)
for log_h, w, ab in ((20, 2, 1), (20, 2, 2)):
    x = rng.integers(0, 0x78000001, (1 << log_h, w), dtype=np.uint32)
    got = p3.host_u32(dft.coset_lde_batch(p3.dev_u32(x), ab, p3.GENERATOR_MONTY, bit_reversed_out=True))
    assert np.array_equal(got, o.coset_lde_batch(x, ab, p3.GENERATOR_MONTY, True)), (log_h, w, ab)
print("k3 check ok")
