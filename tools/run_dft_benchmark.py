#!/usr/bin/env python3
"""The reference's DFT benchmark protocol (native/src/fib_air.rs:98-222: 11 shapes, warmup 1, repeats 10,
avg/median/p95, e2e / e2e-batched / kernel-only views, CPU column + equality check) on the hip backend.
The CPU column is the C oracle (single thread).  Usage on the GPU box:  python tools/run_dft_benchmark.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package  # noqa: E402
from oracle import oracle as o  # noqa: E402  (benchmark tool: the oracle is the CPU column only)

p3 = load_package()
o.build()
text, rows = p3.run_dft_benchmark(cpu_dft=o.dft_batch)
print(text)
