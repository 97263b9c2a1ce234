out=gpurun_out/r04_q4_sweep2.txt
: > $out
for cfg in "0 1" "15 0" "14 0" "15 1" "0 1" "15 0"; do
  set -- $cfg
  echo "== P3HIP_Q4_MAX_LOG=$1 P3HIP_Q4_PRIO=$2" >> $out
  P3HIP_Q4_MAX_LOG=$1 P3HIP_Q4_PRIO=$2 python3 tools/single_proof_latency.py 20 poseidon2 0 >> $out 2>/dev/null
  P3HIP_Q4_MAX_LOG=$1 P3HIP_Q4_PRIO=$2 python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('4-prover bench cfg2: %.1f proofs/s' % d['value'])" >> $out
done
cat $out
