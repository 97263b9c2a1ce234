#!/bin/bash
# A/B on one box: the hiding bench (Keccak, 2^19 rows, four provers) with the integer LDE kernels at 2^20/2^21 rows (default) against
# fp64 butterflies there (P3HIP_NTT_NARROW_F64=7: a third fewer VALU instructions, slower standalone).  Runs the tree of commit 338e019
# (the last one that still has the switch), checked out under _old_tree/ and built beforehand.
set -e
cd _old_tree
out=../gpurun_out/r05_f64_lde_in_hiding_ab.txt
: > $out
for rep in 1 2; do
  for v in default 7; do
    if [ $v = default ]; then unset P3HIP_NTT_NARROW_F64; else export P3HIP_NTT_NARROW_F64=$v; fi
    echo "== rep $rep P3HIP_NTT_NARROW_F64=$v" >> $out
    timeout -k 10 200 python bench.py --hash keccak --hiding --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'])" >> $out
  done
done
cat $out
