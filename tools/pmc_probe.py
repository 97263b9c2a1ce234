#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, MI355X_MICROARCH.md §HBM):
kernels with KNOWN byte counts for calibration + the LDE launches whose HBM traffic we want.
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_probe.py"""
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

# calibration 1: fib_trace 2^24 rows -> writes 2^24 * 8 B = 128 MiB (8 B per lane), reads nothing
t = p3.generate_trace_rows(0, 1, 1 << 24)
torch.cuda.synchronize()
# calibration 2: poseidon2 permute of 2^22 states in place: reads 256 MiB + writes 256 MiB (16 B per lane)
st = torch.randint(0, P, (1 << 22, 16), dtype=torch.int32, device="cuda")
p3._lib.check(L.p3hip_poseidon2_permute_dev(C.c_void_p(st.data_ptr()), 1 << 22, sp()))
torch.cuda.synchronize()
# calibration 3: leaf hash of 2^24 x 2 (reads 128 MiB at 8 B per lane, writes 512 MiB digests) — via commit
mm = p3.MerkleTreeMmcs()
root, tree = mm.commit([t])
tree.free()
torch.cuda.synchronize()
# subject: coset LDE 2^24 x 2, blowup 4 (cfg3): algorithmic 128 MiB in + 512 MiB out; and cfg2 2^20 x 2 blowup 2
dft = p3.GpuDft.with_backend(p3.BackendKind.Hip)
for _ in range(2):
    y = dft.coset_lde_batch(t, 2, p3.GENERATOR_MONTY, bit_reversed_out=True)
torch.cuda.synchronize()
del y
t20 = p3.generate_trace_rows(0, 1, 1 << 20)
for _ in range(3):
    y = dft.coset_lde_batch(t20, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
torch.cuda.synchronize()
del y, t20
# subject: the wide trace of BASELINE configs[4], 2^16 x 2633, blowup 2 (general plans).  Marker launch right before it: the only
# fib_trace of 2^10 rows in this run; the summariser sums every transform launch after the LAST marker.
from plonky3_mobile_amd.fib_air import benchmark_input  # noqa: E402
xw = p3.dev_u32(benchmark_input(1 << 16, 2633))
for _ in range(2):
    p3.generate_trace_rows(0, 1, 1 << 10)
    y = dft.coset_lde_batch(xw, 1, p3.GENERATOR_MONTY, bit_reversed_out=True)
    torch.cuda.synchronize()
print("done")
