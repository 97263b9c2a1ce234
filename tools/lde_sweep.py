#!/usr/bin/env python3
"""coset_lde_batch (bit-reversed output) timing sweep over heights/widths/blowups, for choosing the planner's
thresholds: run once with P3HIP_NTT_NARROW=0 and once with =1 and compare.   python tools/lde_sweep.py"""
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
for w in [int(v) for v in os.environ.get("SWEEP_W", "2,4,8").split(",")]:
    for ab in (1, 2):
        for log_h in range(int(os.environ.get("SWEEP_LO", "14")), int(os.environ.get("SWEEP_HI", "24"))):
            if (1 << (log_h + ab)) * w * 4 > 1 << 31:
                continue
            h = 1 << log_h
            x = torch.randint(0, P, (h, w), dtype=torch.int32, device="cuda")
            y = torch.empty((h << ab, w), dtype=torch.int32, device="cuda")
            run = lambda: p3._lib.check(L.p3hip_coset_lde_batch_bb31_dev(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),  # noqa: E731
                                                                       h, w, ab, p3.GENERATOR_MONTY, 1, sp()))
            run(); run()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                run()
            e.record()
            torch.cuda.synchronize()
            us = s.elapsed_time(e) * 100
            print("w=%d blowup=%d 2^%d: %.1f us  %.0f GB/s" % (w, 1 << ab, log_h, us, 4 * h * w * (1 + (1 << ab)) / us / 1e3))
            del x, y
