#!/bin/bash
# Round 5, VERDICT item 1: the multi-rank code path (RCCL process group, descriptor scatter, proof gather) at ONE rank against the
# plain single-rank run on the same box, alternating.  Output: gpurun_out/r05_dist_ab/*.json (+ a summary).
set -e
out=gpurun_out/r05_dist_ab
mkdir -p $out
for i in 1 2 3; do
  python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 2 > $out/plain_$i.json 2> $out/plain_$i.err
  P3HIP_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-extras --steps 40 --warmup 2 > $out/dist_$i.json 2> $out/dist_$i.err
  echo "pair $i done" >> $out/progress.txt
done
python - <<'PY'
import json, glob
out = "gpurun_out/r05_dist_ab"
rows = []
for kind in ("plain", "dist"):
    for f in sorted(glob.glob("%s/%s_*.json" % (out, kind))):
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["ranks"][0]
        rows.append((kind, f, d["value"], d["steps"], r["scatter_wait_s"], r["gather_wait_s"], r["prover_join_s"], r["prove_wall_s"]))
with open(out + "/summary.txt", "w") as f:
    for r in rows:
        f.write("%-5s %s  %.1f proofs/s  steps %d  scatter_wait %.4f s  gather_wait %.4f s  join %.3f s  wall %.3f s\n" % r)
    pv = [r[2] for r in rows if r[0] == "plain"]; dv = [r[2] for r in rows if r[0] == "dist"]
    f.write("plain mean %.1f  dist mean %.1f  ratio %.4f\n" % (sum(pv) / len(pv), sum(dv) / len(dv), (sum(dv) / len(dv)) / (sum(pv) / len(pv))))
print(open(out + "/summary.txt").read())
PY
