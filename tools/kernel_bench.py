#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels through the C ABI (run on the GPU box):
   python tools/kernel_bench.py [perm] [lde] [commit]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
L = p3._lib.lib()
P = 0x78000001


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def sp():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


which = sys.argv[1:] or ["perm", "lde", "commit"]
if "perm" in which:
    n = 1 << 22
    st = torch.randint(0, P, (n, 16), dtype=torch.int32, device="cuda")
    ms = timeit(lambda: p3._lib.check(L.p3hip_poseidon2_permute_dev(C.c_void_p(st.data_ptr()), n, sp())))
    print("poseidon2 permute: %.1f us for 2^22 states = %.2f Gperm/s" % (ms * 1e3, n / ms / 1e6))
if "lde" in which:
    for log_h, w, ab in [(20, 2, 1), (20, 4, 1), (20, 128, 1), (24, 2, 2), (16, 2633, 1)]:
        h = 1 << log_h
        x = torch.randint(0, P, (h, w), dtype=torch.int32, device="cuda")
        y = torch.empty((h << ab, w), dtype=torch.int32, device="cuda")
        ms = timeit(lambda: p3._lib.check(L.p3hip_coset_lde_batch_bb31_dev(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                                                                          h, w, ab, p3.GENERATOR_MONTY, 1, sp())), 5)
        nbytes = 4 * h * w * (1 + (1 << ab))
        print("coset_lde 2^%d x %d blowup %d: %.1f us, %.1f GB/s algorithmic" % (log_h, w, 1 << ab, ms * 1e3, nbytes / ms / 1e6))
        del x, y
    for log_h, w in [(14, 128), (20, 2), (22, 2)]:
        h = 1 << log_h
        x = torch.randint(0, P, (h, w), dtype=torch.int32, device="cuda")
        y = torch.empty_like(x)
        ms = timeit(lambda: p3._lib.check(L.p3hip_dft_batch_bb31_dev(C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), h, w, sp())), 5)
        print("dft_batch 2^%d x %d: %.1f us, %.1f GB/s algorithmic" % (log_h, w, ms * 1e3, 8 * h * w / ms / 1e6))
if "keccak" in which or not sys.argv[1:]:
    n = 1 << 21
    st = torch.randint(-2**62, 2**62, (n, 25), dtype=torch.int64, device="cuda")
    ms = timeit(lambda: p3._lib.check(L.p3hip_keccak_f_dev(C.c_void_p(st.data_ptr()), n, sp())))
    print("keccak-f[1600]: %.1f us for 2^21 states = %.2f Gperm/s, %.1f GB/s" % (ms * 1e3, n / ms / 1e6, 400 * n / ms / 1e6))
    km = p3.MerkleTreeMmcs(hash="keccak")
    for log_h, w in [(21, 2), (21, 4), (17, 2633)]:
        h = 1 << log_h
        x = torch.randint(0, P, (h, w), dtype=torch.int32, device="cuda")

        def kcommit():
            _, t = km.commit([x])
            t.free()
        ms = timeit(kcommit, 5)
        perms = h * ((((w + 1) // 2) + 16) // 17) + h - 1
        print("keccak mmcs commit 2^%d x %d: %.1f us, %.2f Gperm/s, %.1f GB/s algorithmic" % (log_h, w, ms * 1e3, perms / ms / 1e6, (4 * h * w + 32 * (2 * h - 1)) / ms / 1e6))
        del x
if "commit" in which:
    mm = p3.MerkleTreeMmcs()
    for log_h, w in [(21, 2), (21, 4), (20, 8), (12, 8), (17, 2633)]:
        h = 1 << log_h
        x = torch.randint(0, P, (h, w), dtype=torch.int32, device="cuda")

        def go():
            _, t = mm.commit([x])
            t.free()
        ms = timeit(go, 5)
        perms = h * ((w + 7) // 8) + h - 1
        print("mmcs commit 2^%d x %d: %.1f us, %.2f Gperm/s, %.1f GB/s algorithmic" % (
            log_h, w, ms * 1e3, perms / ms / 1e6, (4 * h * w + 32 * (2 * h - 1)) / ms / 1e6))
        del x
