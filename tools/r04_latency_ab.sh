# Single-prover latency and 4-prover throughput with and without the one-state-per-quad Poseidon2 layers (P3HIP_Q4_MAX_LOG), both hashes
# where it applies; the n = 8 reference instance.  Output: gpurun_out/r04_latency_ab.txt
set -e
out=gpurun_out/r04_latency_ab.txt
: > $out
for q4 in 0 15; do
  echo "== P3HIP_Q4_MAX_LOG=$q4" >> $out
  P3HIP_Q4_MAX_LOG=$q4 python3 tools/single_proof_latency.py 20 poseidon2 0 >> $out 2>/dev/null
  P3HIP_Q4_MAX_LOG=$q4 python3 tools/single_proof_latency.py 20 poseidon2 1 >> $out 2>/dev/null
  P3HIP_Q4_MAX_LOG=$q4 python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4-prover bench cfg2: %.1f proofs/s' % d['value'])" >> $out
done
echo "== independent of the switch" >> $out
python3 tools/single_proof_latency.py 20 keccak 0 >> $out 2>/dev/null
python3 tools/single_proof_latency.py 19 keccak 1 >> $out 2>/dev/null
python3 tools/single_proof_latency.py 3 keccak 1 40 >> $out 2>/dev/null
cat $out
