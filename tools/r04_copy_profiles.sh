#!/bin/bash
# copies the summaries of tools/collect_profiles_r04.sh that are to be judged from gpurun_out/r04_prof into profiles/ (tracked)
S=gpurun_out/r04_prof; D=profiles
cp $S/bench_default.json $D/r04_bench_default_run_with_other_workloads.json
cp $S/bench_cfg3.json $D/r04_bench_cfg3.json; cp $S/bench_cfg5.json $D/r04_bench_cfg5.json; cp $S/bench_cfg5_keccak.json $D/r04_bench_cfg5_keccak.json
cp $S/bench_keccak.json $D/r04_bench_keccak.json; cp $S/bench_keccak_hiding.json $D/r04_bench_keccak_hiding.json
cp $S/bench_cfg4_1gpu.json $D/r04_bench_cfg4_one_gpu.json
cp $S/soak.txt $D/r04_soak.txt
cp $S/prof_lde/lde_kernel_stats.csv $D/r04_lde_unit_kernel_stats.csv; cp $S/prof_lde24/lde24_kernel_stats.csv $D/r04_lde_unit_cfg3_kernel_stats.csv
cp $S/lde_unit_cfg2.json $D/r04_lde_unit_cfg2_under_rocprof.json
head -40 $S/prof_bench/bench_kernel_stats.csv > $D/r04_bench_cfg2_kernel_stats_4_provers.csv
head -40 $S/prof_hid/hid_kernel_stats.csv > $D/r04_bench_keccak_hiding_kernel_stats_4_provers.csv
cp $S/hiding_solo.txt $D/r04_hiding_prover_alone_kernel_times.txt; cp $S/hiding_valu.txt $D/r04_hiding_prover_valu_instructions.txt
cp $S/bench_cfg4_2rank_gloo_rehearsal.json $D/r04_cfg4_2rank_GLOO_rehearsal_one_gpu_via_bench_gpus_2.json
cp $S/bench_1rank_rccl_forced_dist.json $D/r04_bench_1rank_RCCL_forced_dist_path.json
cp $S/hiding_bench.txt $D/r04_hiding_prover_timing.txt
ls -la $D | grep r04_
