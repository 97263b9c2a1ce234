#!/bin/bash
# Same-box A/B of the round-5 FRI tail for the Keccak hashes and the hiding provers (fri_tail_any_kernel, prover_wg1.hip.inc): the product
# library against a diagnostic build with -DFRI_TAIL_ANY_OFF=1 (a launch per step, as before), one proof at a time, latency profile.
# RUN AT COMMIT ce13ca9 (the tree that has fri_tail_any_kernel and the macro): the tail measured neutral to slower and the next commit took it out;
# on any later tree both libraries are the same code.  Record: profiles/r05_fri_tail_any_ab.txt.
set -e
( cd plonky3-mobile_amd/csrc
  FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
  mkdir -p ../../tools/_bin
  /opt/rocm/bin/hipcc $FLAGS -DFRI_TAIL_ANY_OFF=1 -c prover.hip -o ../../tools/_bin/prover_notail.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_bin/libp3hip_notail.so _obj/context.o _obj/ntt.o _obj/mmcs.o _obj/fib_air.o _obj/c_api.o ../../tools/_bin/prover_notail.o _obj/verifier.o _obj/rng.o _obj/front_end.o )
out=gpurun_out/r05_fri_tail_any_ab.txt
: > $out
for rep in 1 2 3; do
  for lib in product notail; do
    if [ $lib = product ]; then unset P3HIP_LIB; else export P3HIP_LIB=$PWD/tools/_bin/libp3hip_notail.so; fi
    echo "== rep $rep $lib" >> $out
    python tools/single_proof_latency.py 19 keccak 1 24 latency >> $out
    python tools/single_proof_latency.py 20 keccak 0 24 latency >> $out
    python tools/single_proof_latency.py 19 poseidon2 1 24 latency >> $out
    python tools/single_proof_latency.py 10 keccak 1 40 latency >> $out
  done
done
cat $out
