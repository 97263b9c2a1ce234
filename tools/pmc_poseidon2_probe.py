#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC pass that backs bench.py's `valu_roofline` (profiles/r02_pmc_poseidon2.json):
the Poseidon2 kernels the prover runs, at the prover's sizes, plus a calibration kernel of known VALU occupancy.
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
            SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_p2 -- \
            python3 tools/pmc_poseidon2_probe.py
Launches, in order: raw permute of 2^22 states (poseidon2_permute_f64_kernel: 1 permutation per lane, pure VALU between
one load and one store per lane — the calibration of "VALU busy" for this instruction mix); Merkle commit of 2^21 x 2
(the trace tree of cfg2: leaf_hash_f64 2^21, compress_layer_f64 2^20..2^15, tree_levels_coop below); commit of
2^20 x 8 (FRI round 0); commit of 2^26 x 2 (the trace tree of cfg3)."""
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
mm = p3.MerkleTreeMmcs()
for rep in range(2):  # first pass warms tables / clocks; the summariser takes the LAST launch of each kernel shape
    st = torch.randint(0, P, (1 << 22, 16), dtype=torch.int32, device="cuda")
    p3._lib.check(L.p3hip_poseidon2_permute_dev(C.c_void_p(st.data_ptr()), 1 << 22, sp()))
    torch.cuda.synchronize()
    del st
    for log_h, w in ((21, 2), (20, 8), (26, 2)):
        x = torch.randint(0, P, (1 << log_h, w), dtype=torch.int32, device="cuda")
        root, tree = mm.commit([x])
        torch.cuda.synchronize()
        tree.free()
        del x
print("done")
