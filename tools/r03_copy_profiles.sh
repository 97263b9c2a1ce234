#!/bin/bash
# copies the judged summaries of gpurun_out/r03_prof (tools/collect_profiles_r03.sh) into profiles/
set -e
S=gpurun_out/r03_prof
cp $S/r03_pmc_poseidon2.json $S/r03_pmc_keccak.json $S/r03_pmc_lde.json profiles/   # written on the GPU box, brought back under gpurun_out
cp $S/prof_lde/lde_kernel_stats.csv profiles/r03_lde_unit_kernel_stats.csv
cp $S/prof_lde24/lde24_kernel_stats.csv profiles/r03_lde_unit_cfg3_kernel_stats.csv
cp $S/prof_bench/bench_kernel_stats.csv profiles/r03_bench_kernel_stats.csv
cp $S/prof_bench_1t/bench1_kernel_stats.csv profiles/r03_bench_kernel_stats_single_prover.csv
for f in cfg2 cfg3 cfg5 cfg4_1gpu keccak keccak_hiding; do tail -n 1 $S/bench_$f.json > profiles/r03_bench_$f.json; done
tail -n 1 $S/bench_cfg4_2rank_gloo_rehearsal.json > profiles/r03_bench_cfg4_2rank_GLOO_rehearsal_one_gpu.json
grep -v amdgpu.ids $S/hiding_bench.txt > profiles/r03_hiding_prover_timing.txt
cp $S/lde_unit_cfg2.json profiles/r03_lde_unit_cfg2_under_rocprof.json
ls profiles/r03_*
