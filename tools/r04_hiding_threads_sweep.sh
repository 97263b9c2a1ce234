# bench.py --hash keccak --hiding with 3..8 concurrent provers (does more overlap of the memory-bound phases with the hash layers pay?)
set -e
for t in 3 4 5 6 8; do
  python bench.py --hash keccak --hiding --no-cpu-baseline --no-extras --threads $t --steps 8 --warmup 2 > gpurun_out/r04_hid_threads_$t.json 2> gpurun_out/r04_hid_threads_$t.err
  python3 -c "
import json,sys; d=json.loads(open('gpurun_out/r04_hid_threads_$t.json').read().strip().splitlines()[-1]); print('threads $t: %.1f proofs/s' % d['value'])"
done | tee gpurun_out/r04_hid_threads_sweep.txt
