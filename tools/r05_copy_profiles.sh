#!/bin/bash
# copies the summaries of tools/collect_profiles_r05.sh that are to be judged from gpurun_out/ into profiles/ (tracked)
S=gpurun_out/r05_prof; D=profiles
cp gpurun_out/r05_pmc_proofs.json $D/r05_pmc_proofs.json; cp gpurun_out/r05_pmc_lde_valu.json $D/r05_pmc_lde_valu.json; cp gpurun_out/r05_pmc_lde.json $D/r05_pmc_lde.json
cp $S/pmc_proofs.txt $D/r05_pmc_proofs_summary.txt
cp $S/bench_default.json $D/r05_bench_default_run_with_other_workloads.json
cp $S/bench_cfg3.json $D/r05_bench_cfg3.json; cp $S/bench_cfg5.json $D/r05_bench_cfg5.json
cp $S/bench_keccak.json $D/r05_bench_keccak.json; cp $S/bench_keccak_hiding.json $D/r05_bench_keccak_hiding.json
cp $S/bench_cfg4_1gpu.json $D/r05_bench_cfg4_one_gpu.json
cp $S/soak.txt $D/r05_soak.txt
cp $S/prof_lde/lde_kernel_stats.csv $D/r05_lde_unit_kernel_stats.csv; cp $S/prof_lde24/lde24_kernel_stats.csv $D/r05_lde_unit_cfg3_kernel_stats.csv
cp $S/lde_unit_cfg2.json $D/r05_lde_unit_cfg2_under_rocprof.json
head -40 $S/prof_bench/bench_kernel_stats.csv > $D/r05_bench_cfg2_kernel_stats_4_provers.csv
head -40 $S/prof_hid/hid_kernel_stats.csv > $D/r05_bench_keccak_hiding_kernel_stats_4_provers.csv
cp $S/bench_cfg4_2rank_gloo_rehearsal.json $D/r05_cfg4_2rank_GLOO_rehearsal_one_gpu_via_bench_gpus_2.json
ls -la $D | grep r05_
