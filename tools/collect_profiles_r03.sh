#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root:  bash tools/collect_profiles_r03.sh
# Everything lands under gpurun_out/r03_prof/; tools/r03_copy_profiles.sh copies what is to be judged into profiles/.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r03_prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
say() { echo "== $* ($(date +%T))"; }
CTRS="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
say "PMC: Poseidon2 + calibration, Keccak"
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_p2_cal -- $ROOT/tools/_bin/clock_probe > $OUT/pmc_p2_cal.log 2>&1
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_p2 -- python3 $ROOT/tools/pmc_poseidon2_probe.py > $OUT/pmc_p2.log 2>&1
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pmc_kk -- python3 $ROOT/tools/pmc_keccak_probe.py > $OUT/pmc_kk.log 2>&1
say "PMC: LDE traffic (FETCH_SIZE, WRITE_SIZE separately)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/pmc_probe.py > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/pmc_probe.py > $OUT/pmc_write.log 2>&1
cd $ROOT
python3 tools/pmc_poseidon2_summarize.py $OUT/pmc_p2 $OUT/pmc_p2_cal profiles/r03_pmc_poseidon2.json > $OUT/pmc_p2_summary.txt 2>&1
python3 tools/pmc_keccak_summarize.py $OUT/pmc_kk profiles/r03_pmc_keccak.json > $OUT/pmc_kk_summary.txt 2>&1
python3 tools/pmc_summarize.py $OUT/pmc_fetch $OUT/pmc_write profiles/r03_pmc_lde.json > $OUT/pmc_lde_summary.txt 2>&1
cp profiles/r03_pmc_poseidon2.json profiles/r03_pmc_keccak.json profiles/r03_pmc_lde.json $OUT/
cd /tmp
say "LDE unit under rocprofv3 --kernel-trace --stats (the same command bench.py's roofline measures)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lde -o lde -- python3 $ROOT/tools/lde_unit_profile.py 20 1 > $OUT/lde_unit_cfg2.json 2> $OUT/prof_lde.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lde24 -o lde24 -- python3 $ROOT/tools/lde_unit_profile.py 24 2 > $OUT/lde_unit_cfg3.json 2> $OUT/prof_lde24.err
say "bench cfg2 under rocprofv3 --kernel-trace --stats: 4 provers, then a single prover"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_cfg2_under_rocprof.json 2> $OUT/prof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench_1t -o bench1 -- python3 $ROOT/bench.py --steps 4 --warmup 1 --threads 1 --batch 8 --no-cpu-baseline --no-extras > $OUT/bench_cfg2_1prover_under_rocprof.json 2> $OUT/prof_bench_1t.err
cd $ROOT
say "bench lines (un-profiled)"
python3 bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python3 bench.py --workload cfg3 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --workload cfg5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python3 bench.py --workload cfg4 --no-cpu-baseline --no-extras > $OUT/bench_cfg4_1gpu.json 2> $OUT/bench_cfg4.err
python3 bench.py --hash keccak > $OUT/bench_keccak.json 2> $OUT/bench_keccak.err
python3 bench.py --hash keccak --hiding > $OUT/bench_keccak_hiding.json 2> $OUT/bench_keccak_hiding.err
say "2 ranks started by bench.py --gpus 2 over gloo on one GPU (NOT RCCL)"
P3HIP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --workload cfg4 --steps 5 --warmup 1 > $OUT/bench_cfg4_2rank_gloo_rehearsal.json 2> $OUT/bench_2rank_gloo.err
say "hiding prover timing"
python3 tools/hiding_bench.py > $OUT/hiding_bench.txt 2>&1
say done
