"""How many OpenMP threads the oracle prover should get on this host: one 2^20-row proof per thread count.
On a one-GPU box (16-core share, 256 logical cores visible): 16 -> 1.7 s, 32 -> 1.7, 64 -> 2.1, 128 -> 2.8, 256 -> 35 s."""
import sys, time
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from oracle import oracle as o
o.build()
fp = o.FriParams(1, 0, 100, 16)
print("max", o.max_threads())
for th in (16, 32, 64, 128, 256):
    o.set_threads(th); t = time.time(); o.prove_fib_air(0, 1, 20, fp); print(th, round(time.time() - t, 2), flush=True)
