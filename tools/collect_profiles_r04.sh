#!/bin/bash
# Round-4 evidence, run on the GPU box from the repo root:  bash tools/collect_profiles_r04.sh
# Everything lands under gpurun_out/r04_prof/; tools/r04_copy_profiles.sh copies what is to be judged into profiles/.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r04_prof
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
say() { echo "== $* ($(date +%T))"; }
say "bench lines (un-profiled)"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 bench.py --workload cfg3 --no-extras > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python3 bench.py --workload cfg5 --no-extras > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python3 bench.py --workload cfg5 --hash keccak --no-extras --no-cpu-baseline > $OUT/bench_cfg5_keccak.json 2> $OUT/bench_cfg5_keccak.err
python3 bench.py --hash keccak --no-extras > $OUT/bench_keccak.json 2> $OUT/bench_keccak.err
python3 bench.py --hash keccak --hiding --no-extras > $OUT/bench_keccak_hiding.json 2> $OUT/bench_keccak_hiding.err
python3 bench.py --workload cfg4 --no-cpu-baseline --no-extras > $OUT/bench_cfg4_1gpu.json 2> $OUT/bench_cfg4.err
say "soak: cfg2 x3, hiding with 4 and 5 provers x3"
for i in 1 2 3; do
  python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 run $i: %.1f proofs/s' % d['value'])"
  for t in 4 5; do
    python3 bench.py --hash keccak --hiding --no-cpu-baseline --no-extras --threads $t --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keccak + hiding, $t provers, run $i: %.1f proofs/s' % d['value'])"
  done
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
say "hiding prover alone: kernel times and VALU instructions per proof"
bash tools/r04_hiding_solo_stats.sh r04_prof/hiding_solo > /dev/null 2>&1
bash tools/r04_hiding_valu_share.sh r04_prof/hiding_valu > /dev/null 2>&1
say "2 ranks started by bench.py --gpus 2 over gloo on one GPU (NOT RCCL): the self-explaining N-rank line"
P3HIP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --workload cfg4 --steps 5 --warmup 1 2> $OUT/bench_2rank_gloo.err | grep '^{' > $OUT/bench_cfg4_2rank_gloo_rehearsal.json
say "one rank on RCCL through the N > 1 code path"
P3HIP_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 2> $OUT/bench_1rank_rccl.err | grep '^{' > $OUT/bench_1rank_rccl_forced_dist.json
say "hiding prover timing"
python3 tools/hiding_bench.py > $OUT/hiding_bench.txt 2>&1
say done
