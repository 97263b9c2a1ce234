#!/bin/bash
# Same-box A/B of kk::f_coop with the padding masks and iota folded into v_bitop3 (46 instead of 60 VALU instructions per round) against the
# previous form (tools/_bin/libp3hip_old_fcoop.so, built beforehand from the previous keccak.hip.h), one proof at a time, latency profile.
# Result (profiles/r05_fcoop_fewer_instructions_ab.txt): no change — the permutation is bound by its dependent chain — so the tree keeps the old form and
# this script only documents how the pair was measured.
set -e
out=gpurun_out/r05_fcoop_ab.txt
: > $out
for rep in 1 2 3; do
  for lib in new old; do
    if [ $lib = new ]; then unset P3HIP_LIB; else export P3HIP_LIB=$PWD/tools/_bin/libp3hip_old_fcoop.so; fi
    echo "== rep $rep $lib" >> $out
    python tools/single_proof_latency.py 19 keccak 1 24 latency >> $out
    python tools/single_proof_latency.py 20 keccak 0 24 latency >> $out
    python tools/single_proof_latency.py 3 keccak 1 60 latency >> $out
    python tools/single_proof_latency.py 10 keccak 1 40 latency >> $out
  done
done
cat $out
