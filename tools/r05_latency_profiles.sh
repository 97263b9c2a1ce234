#!/bin/bash
# Round 5, VERDICT item 4: single-proof latency under the two creation-time profiles, no environment variable set.
set -e
out=gpurun_out/r05_latency_profiles.txt
: > $out
for rep in 1 2; do
  for prof in latency throughput; do
    python tools/single_proof_latency.py 20 poseidon2 0 16 $prof >> $out
    python tools/single_proof_latency.py 19 keccak 1 16 $prof >> $out
    python tools/single_proof_latency.py 20 keccak 0 16 $prof >> $out
    python tools/single_proof_latency.py 3 keccak 1 40 $prof >> $out
    python tools/single_proof_latency.py 10 poseidon2 0 30 $prof >> $out
  done
done
cat $out
