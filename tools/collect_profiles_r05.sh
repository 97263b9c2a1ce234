#!/bin/bash
# Round-5 evidence, run on the GPU box from the repo root:  bash tools/collect_profiles_r05.sh
# Everything lands under gpurun_out/r05_prof/ (+ the PMC summaries gpurun_out/r05_pmc_*.json); tools/r05_copy_profiles.sh copies what
# is to be judged into profiles/.  The PMC passes run FIRST and their summaries are copied into profiles/ on the box, so that the
# bench lines produced afterwards carry figures of THIS build (bench_support._profile_json compares lib_sha256).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r05_prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
say() { echo "== $* ($(date +%T))"; }
say "PMC: VALU wave-instructions of whole proofs (4 workloads)"
bash tools/r05_pmc_proofs.sh > $OUT/pmc_proofs.txt 2>&1; tail -6 $OUT/pmc_proofs.txt
cp gpurun_out/r05_pmc_proofs.json profiles/r05_pmc_proofs.json
say "PMC: VALU instructions of the LDE unit (cfg2 / cfg3 / cfg5)"
bash tools/r05_lde_valu.sh > $OUT/pmc_lde_valu.txt 2>&1; tail -4 $OUT/pmc_lde_valu.txt
cp gpurun_out/r05_pmc_lde_valu.json profiles/r05_pmc_lde_valu.json
say "PMC: HBM traffic of the LDE unit (FETCH_SIZE and WRITE_SIZE in separate passes)"
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/pmc_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/pmc_probe.py > $OUT/pmc_write.log 2>&1
cd $ROOT
python3 tools/pmc_summarize.py $OUT/pmc_fetch $OUT/pmc_write gpurun_out/r05_pmc_lde.json > $OUT/pmc_lde.txt 2>&1; cat $OUT/pmc_lde.txt
cp gpurun_out/r05_pmc_lde.json profiles/r05_pmc_lde.json
say "bench lines (un-profiled)"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --workload cfg3 --no-extras > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --workload cfg5 --no-extras > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python3 bench.py --hash keccak --no-extras > $OUT/bench_keccak.json 2> $OUT/bench_keccak.err
python3 bench.py --hash keccak --hiding --no-extras > $OUT/bench_keccak_hiding.json 2> $OUT/bench_keccak_hiding.err
python3 bench.py --workload cfg4 --no-cpu-baseline --no-extras > $OUT/bench_cfg4_1gpu.json 2> $OUT/bench_cfg4.err
say "soak: cfg2 x3, hiding x3"
for i in 1 2 3; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 run $i: %.1f proofs/s' % d['value'])"
  python3 bench.py --hash keccak --hiding --no-cpu-baseline --no-extras --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keccak + hiding run $i: %.1f proofs/s' % d['value'])"
done > $OUT/soak.txt 2>&1
cat $OUT/soak.txt
cd /tmp
say "LDE unit under rocprofv3 --kernel-trace --stats (the same command bench.py's roofline measures)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lde -o lde -- python3 $ROOT/tools/lde_unit_profile.py 20 1 > $OUT/lde_unit_cfg2.json 2> $OUT/prof_lde.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lde24 -o lde24 -- python3 $ROOT/tools/lde_unit_profile.py 24 2 > $OUT/lde_unit_cfg3.json 2> $OUT/prof_lde24.err
say "bench cfg2 and the hiding configuration under rocprofv3 --kernel-trace --stats (4 provers)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_cfg2_under_rocprof.json 2> $OUT/prof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hid -o hid -- python3 $ROOT/bench.py --hash keccak --hiding --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_hiding_under_rocprof.json 2> $OUT/prof_hid.err
cd $ROOT
say "2 ranks started by bench.py --gpus 2 over gloo on one GPU (NOT RCCL): a rehearsal of the launcher and the collectives"
P3HIP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --workload cfg4 --steps 5 --warmup 1 2> $OUT/bench_2rank_gloo.err | grep '^{' > $OUT/bench_cfg4_2rank_gloo_rehearsal.json
say "single-proof latency under the two creation-time profiles (no environment variable)"
bash tools/r05_latency_profiles.sh > $OUT/latency.log 2>&1; cp gpurun_out/r05_latency_profiles.txt $OUT/latency_profiles.txt
say "the one-launch prover: phase stamps of a diagnostic build (tools/r05_tiny_stamps.sh built it into tools/_bin before the run)"
if [ -f tools/_bin/libp3hip_stamps.so ]; then
  P3HIP_LIB=$ROOT/tools/_bin/libp3hip_stamps.so python3 tools/single_proof_latency.py 3 keccak 1 4 latency > $OUT/tiny_stamps_keccak.txt 2>&1
  P3HIP_LIB=$ROOT/tools/_bin/libp3hip_stamps.so python3 tools/single_proof_latency.py 3 poseidon2 1 4 latency > $OUT/tiny_stamps_poseidon2.txt 2>&1
fi
say done
