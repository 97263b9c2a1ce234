#!/usr/bin/env python3
"""The exact measurement bench.py reports as `roofline` (FibAirJob.lde_roofline: the coset LDE of one 2^h x 2 trace,
HIP events on the launch stream), run alone so that a `rocprofv3 --kernel-trace --stats` of this command shows the
same launches and nothing else:
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lde_unit -o lde -- python3 tools/lde_unit_profile.py [log_h] [log_blowup]
The sum of the three kernels' average durations must agree with the printed avg_us up to the launch gaps."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
from __graft_entry__ import load_package  # noqa: E402

p3 = load_package()
from importlib import import_module  # noqa: E402

bs = import_module("plonky3_mobile_amd.bench_support")
log_h = int(sys.argv[1]) if len(sys.argv) > 1 else 20
log_b = int(sys.argv[2]) if len(sys.argv) > 2 else 1


class _Unit(bs.FibAirJob):
    def __init__(self):  # no prover workers: only the LDE measurement
        self.device = torch.cuda.current_device()
        self.log_height, self.log_blowup, self.n = log_h, log_b, 1 << log_h
        self.threads = 1

    def _concurrent_lde(self, nbytes, reps=40):
        return None


print(json.dumps(_Unit().lde_roofline(reps=20)))
