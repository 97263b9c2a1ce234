#!/bin/bash
# Same-box A/B of the one-launch prover with 1024 threads (tools/_bin/libp3hip_tiny1k.so: prover.hip compiled from a copy of csrc with
# TINY_THREADS = 1024, linked with the product's other objects) against the product's 512.  Result: slower (profiles/r05_tiny_1024_threads_ab.txt).
set -e
out=gpurun_out/r05_tiny_1024_threads_ab.txt
: > $out
P3HIP_LIB=$PWD/tools/_bin/libp3hip_tiny1k.so timeout -k 10 200 python -m pytest tests/test_gpu_hiding.py -x -q -m gpu -k "one_launch" 2>&1 | tail -2 >> $out
for rep in 1 2 3; do
  for lib in product tiny1k; do
    if [ $lib = product ]; then unset P3HIP_LIB; else export P3HIP_LIB=$PWD/tools/_bin/libp3hip_tiny1k.so; fi
    echo "== rep $rep $lib" >> $out
    python tools/single_proof_latency.py 3 keccak 1 80 latency >> $out
    python tools/single_proof_latency.py 3 poseidon2 1 80 latency >> $out
    python tools/single_proof_latency.py 5 keccak 1 60 latency >> $out
  done
done
cat $out
