#!/usr/bin/env python3
"""Do concurrent provers cost the hash layers anything by themselves?  N host threads, each with its own stream and context, commit a
2^log_h x w matrix `reps` times (Keccak or Poseidon2 MMCS); aggregate permutations/s against one thread alone.
   python3 tools/concurrent_commit_probe.py [hash=keccak] [log_h=22] [w=8] [reps=6]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
hash = sys.argv[1] if len(sys.argv) > 1 else "keccak"
log_h = int(sys.argv[2]) if len(sys.argv) > 2 else 22
w = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 6
P = 0x78000001
h = 1 << log_h
perms = h * ((w + 7) // 8 if hash == "poseidon2" else ((w + 1) // 2 + 16) // 17) + h - 1


import ctypes as C  # noqa: E402

L = p3._lib.lib()
kind = 1 if hash == "keccak" else 0


def worker(mat, out, barrier):
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        # commit into pre-allocated layers: nothing is allocated, nothing synchronises (what a prover does)
        layers = torch.empty((L.p3hip_mmcs_layer_words(h),), dtype=torch.int32, device="cuda")
        ptrs, hs, ws = (C.c_void_p * 1)(mat.data_ptr()), (C.c_size_t * 1)(h), (C.c_size_t * 1)(w)
        sp = C.c_void_p(s.cuda_stream)

        def commit():
            t = C.c_void_p()
            p3._lib.check(L.p3hip_mmcs_commit_into_dev(kind, ptrs, hs, ws, 1, C.c_void_p(layers.data_ptr()), C.byref(t), sp))
            L.p3hip_mmcs_free(t)
        commit()
        s.synchronize()
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(reps):
            commit()
        s.synchronize()
        out.append(time.perf_counter() - t0)
        L.p3hip_release_thread_context()


for n in (1, 2, 4, 1, 4):
    mats = [torch.randint(0, P, (h, w), dtype=torch.int32, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    out, barrier = [], threading.Barrier(n)
    ts = [threading.Thread(target=worker, args=(m, out, barrier)) for m in mats]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    wall = max(out)
    print("%d concurrent committers, %s, 2^%d x %d: %.2f Gperm/s aggregate (%.2f ms per commit and thread)" % (n, hash, log_h, w, n * reps * perms / wall / 1e9, wall / reps * 1e3))
    del mats
