# one-state-per-quad Poseidon2 layers: threshold sweep, single-proof latency vs 4-prover throughput (two bench runs per setting)
out=gpurun_out/r04_q4_sweep.txt
: > $out
for q4 in 0 13 14 15 0 13 14 15; do
  echo "== P3HIP_Q4_MAX_LOG=$q4" >> $out
  P3HIP_Q4_MAX_LOG=$q4 python3 tools/single_proof_latency.py 20 poseidon2 0 >> $out 2>/dev/null
  P3HIP_Q4_MAX_LOG=$q4 python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4-prover bench cfg2: %.1f proofs/s' % d['value'])" >> $out
done
cat $out
