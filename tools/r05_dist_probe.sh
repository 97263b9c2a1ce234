#!/bin/bash
# What the N > 1 code path still costs a rank (one rank, RCCL): plain / dist / dist without the gather / both with SDMA off
set -e
out=gpurun_out/r05_dist_probe
mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 2 > $out/$name.json 2> $out/$name.err
  python - $out/$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["ranks"][0]
print("%-28s %.1f proofs/s  wall %.3f  join %.3f  scatter %.4f  gather %.4f" % (sys.argv[2], d["value"], r["prove_wall_s"], r["prover_join_s"], r["scatter_wait_s"], r["gather_wait_s"]))
PY
}
for i in 1 2; do
  run plain_$i P3HIP_X=0
  run dist_$i P3HIP_BENCH_FORCE_DIST=1
  run dist_nogather_$i P3HIP_BENCH_FORCE_DIST=1 P3HIP_BENCH_NO_GATHER=1
  run plain_nosdma_$i HSA_ENABLE_SDMA=0
  run dist_nosdma_$i P3HIP_BENCH_FORCE_DIST=1 HSA_ENABLE_SDMA=0
done
