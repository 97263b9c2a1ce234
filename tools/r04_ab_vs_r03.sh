# Same-box A/B of the round-3 build (git ea4e022, checked out and built under tools/_bin/r03tree by the builder) against this tree:
# bench.py of EACH tree with its own library, alternating, three runs each per workload.  Output: gpurun_out/r04_ab_vs_r03.txt
out=gpurun_out/r04_ab_vs_r03.txt
: > $out
R3=tools/_bin/r03tree
run() {  # tree label args...
  tree=$1; label=$2; shift 2
  ( cd $tree && python3 bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null ) | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$label: %.1f %s' % (d['value'], d['unit']))" >> $out
}
for i in 1 2 3; do
  run $R3 "round 3 cfg2 run $i" --steps 20 --warmup 2
  run .   "round 4 cfg2 run $i" --steps 20 --warmup 2
  run $R3 "round 3 keccak+hiding run $i" --hash keccak --hiding --steps 10 --warmup 2
  run .   "round 4 keccak+hiding run $i" --hash keccak --hiding --steps 10 --warmup 2
done
run $R3 "round 3 keccak" --hash keccak --steps 10 --warmup 2
run .   "round 4 keccak" --hash keccak --steps 10 --warmup 2
run $R3 "round 3 cfg3" --workload cfg3
run .   "round 4 cfg3" --workload cfg3
run $R3 "round 3 cfg5" --workload cfg5
run .   "round 4 cfg5" --workload cfg5
cat $out
