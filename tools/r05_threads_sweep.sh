#!/bin/bash
# Concurrent provers per GPU (host threads / streams), same box: does anything but 4 issue more of the chip's VALU slots?
set -e
out=gpurun_out/r05_threads_sweep.txt
: > $out
for t in 3 4 5 6 8; do
  for wl in "" "--hash keccak --hiding"; do
    echo "== threads $t $wl" >> $out
    timeout -k 10 200 python bench.py --threads $t $wl --no-cpu-baseline --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['unit'], 'valu', d.get('valu_roofline',{}).get('frac'))" >> $out
  done
done
cat $out
