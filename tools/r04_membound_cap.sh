# Experiment: cap the resident workgroups of the hiding prover's streaming kernels (dynamic LDS reserved, never touched) so that other
# provers' hash layers keep the wave slots.  bench.py --hash keccak --hiding, two runs per setting.
out=gpurun_out/r04_membound_cap.txt
: > $out
for lds in 0 40000 80000 0 40000 80000; do
  P3HIP_MEMBOUND_LDS=$lds python3 bench.py --hash keccak --hiding --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('P3HIP_MEMBOUND_LDS=$lds: %.1f proofs/s' % d['value'])" >> $out
done
cat $out
