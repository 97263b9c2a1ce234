#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root:  bash tools/collect_profiles_r02.sh
# Everything lands under gpurun_out/r02_prof/; copy what is to be judged into profiles/ afterwards.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r02_prof
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
say() { echo "== $* ($(date +%T))"; }
say "bench cfg2 (un-profiled)"
python3 $ROOT/bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
say "bench cfg2 under rocprofv3 --kernel-trace --stats"
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_cfg2_under_rocprof.json 2> $OUT/prof_bench.err
say "single prover under rocprofv3"
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench_1t -o bench1 -- python3 $ROOT/bench.py --steps 4 --warmup 1 --threads 1 --batch 8 --no-cpu-baseline > $OUT/bench_cfg2_1prover_under_rocprof.json 2> $OUT/prof_bench_1t.err
say "LDE unit under rocprofv3"
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lde -o lde -- python3 $ROOT/tools/lde_unit_profile.py 20 1 > $OUT/lde_unit_cfg2.json 2> $OUT/prof_lde.err
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lde24 -o lde24 -- python3 $ROOT/tools/lde_unit_profile.py 24 2 > $OUT/lde_unit_cfg3.json 2> $OUT/prof_lde24.err
say "PMC passes of the LDE unit (FETCH_SIZE, WRITE_SIZE separately)"
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/tools/pmc_probe.py > $OUT/pmc_fetch.log 2>&1
PYTHONPATH=$ROOT rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/tools/pmc_probe.py > $OUT/pmc_write.log 2>&1
cd $ROOT
python3 tools/pmc_summarize.py $OUT/pmc_fetch $OUT/pmc_write $OUT/r02_pmc_lde.json > $OUT/pmc_lde_summary.txt 2>&1
say "other workloads"
python3 bench.py --workload cfg3 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --workload cfg5 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python3 bench.py --hash keccak > $OUT/bench_keccak.json 2> $OUT/bench_keccak.err
python3 bench.py --hash keccak --hiding > $OUT/bench_keccak_hiding.json 2> $OUT/bench_keccak_hiding.err
say "2-rank rehearsal over gloo on one GPU (NOT RCCL)"
P3HIP_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --batch 8 --threads 4 > $OUT/bench_2rank_gloo_rehearsal.json 2> $OUT/bench_2rank_gloo.err
say "hiding prover timing"
python3 tools/hiding_bench.py > $OUT/hiding_bench.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head
say done
