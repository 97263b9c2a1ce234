#!/bin/bash
# Phase timeline of the one-launch prover (prover_tiny.hip.inc) at the reference's own instance: a diagnostic library with
# -DTINY_STAMPS=1 (thread 0 prints the 100 MHz wall clock at every phase boundary) next to the product build.
set -e
cd plonky3-mobile_amd/csrc
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
mkdir -p ../../tools/_bin
/opt/rocm/bin/hipcc $FLAGS -DTINY_STAMPS=1 -c prover.hip -o ../../tools/_bin/prover_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_bin/libp3hip_stamps.so _obj/context.o _obj/ntt.o _obj/mmcs.o _obj/fib_air.o _obj/c_api.o ../../tools/_bin/prover_stamps.o _obj/verifier.o _obj/rng.o _obj/front_end.o
echo built tools/_bin/libp3hip_stamps.so
