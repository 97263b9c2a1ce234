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
cp $S/latency_profiles.txt $D/r05_latency_profiles.txt
(echo "# The reference's own instance (n = 8, x = 21, Keccak hashes, hiding MMCS / PCS seeded with 1, FRI (2, 2, 2, 1); native/src/fib_air.rs:56-72) and its"
 echo "# Poseidon2 twin: one proof at a time, tools/single_proof_latency.py 3 <hash> 1 <reps> <profile>.  latency profile = ONE launch (prover_tiny.hip.inc),"
 echo "# throughput profile = the multi-launch sequence (58 launches)."
 grep "2^3 rows" $S/latency_profiles.txt
 echo
 echo "# phase stamps of the one-launch prover (diagnostic build with -DTINY_STAMPS=1, tools/r05_tiny_stamps.sh): thread 0 reads the 100 MHz wall clock at"
 echo "# every phase boundary; '+a at b' = a tenths of a microsecond in the phase, b since kernel entry.  Last of four proofs, Keccak then Poseidon2:"
 grep "^tiny " $S/tiny_stamps_keccak.txt | tail -15
 echo
 grep "^tiny " $S/tiny_stamps_poseidon2.txt | tail -15) > $D/r05_tiny_instance.txt
ls -la $D | grep r05_
