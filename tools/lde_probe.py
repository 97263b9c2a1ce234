#!/usr/bin/env python3
"""A few coset LDEs (bit-reversed output) of given shapes, for rocprofv3 kernel traces / PMC passes of the narrow plan:
   rocprofv3 --kernel-trace --stats ... -- python3 tools/lde_probe.py 20:2:1 22:4:2 [reps]
shape = log_height:width:log_blowup."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
L = p3._lib.lib()
P = 0x78000001
sp = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
shapes = [tuple(int(v) for v in s.split(":")) for s in sys.argv[1:] if ":" in s]
reps = next((int(s) for s in sys.argv[1:] if ":" not in s), 5)
for log_h, w, ab in shapes:
    h = 1 << log_h
    x = torch.randint(0, P, (h, w), dtype=torch.int32, device="cuda")
    y = torch.empty((h << ab, w), dtype=torch.int32, device="cuda")
    for _ in range(reps):
        p3._lib.check(L.p3hip_coset_lde_batch_bb31_dev(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), h, w, ab,
                                                       p3.GENERATOR_MONTY, 1, sp()))
    torch.cuda.synchronize()
    del x, y
print("done")
