#!/bin/bash
# Shader clock granted while the four-prover bench runs: tools/_bin/clock_probe (a 64-thread kernel reading s_memtime against the 100 MHz wall clock)
# from a second process during the timed region.
out=gpurun_out/r05_clock_under_bench.txt
: > $out
for wl in "" "--hash keccak --hiding"; do
  python bench.py $wl --no-cpu-baseline --no-extras --steps 400 --warmup 3 > gpurun_out/clk_bench.json 2>/dev/null &
  pid=$!
  sleep 14
  echo "== during bench.py $wl" >> $out
  for i in 1 2 3; do tools/_bin/clock_probe 2>/dev/null | head -4 | tail -1 >> $out; done
  wait $pid
  python -c "import json; d=json.loads(open('gpurun_out/clk_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['unit'], 'steps', d['steps'], 'ms/step', round(d['ms_per_step'],1))" >> $out
done
cat $out
