#!/bin/bash
# A/B of north_star's "wavefront-shuffle twiddle broadcast" against the LDS broadcast reads the kernels use
# (NARROW_TW_SHUFFLE, ntt_narrow.hip.h), on the coset LDE of the fib_air trace (narrow_fwd2_kernel<10,..> and its two
# companions).  Build of the variant library: every csrc/*.hip with -DNARROW_TW_SHUFFLE=1 -> tools/_bin/libp3hip_twshuffle.so.
# Run on the GPU box from the repo root; writes gpurun_out/r02_twiddle_ab.txt.
ROOT=$(pwd); export TMPDIR=/tmp; OUT=$ROOT/gpurun_out
{
for lib in default twshuffle; do
  if [ $lib = default ]; then unset P3HIP_LIB; else export P3HIP_LIB=$ROOT/tools/_bin/libp3hip_twshuffle.so; fi
  echo "== library: $lib (${P3HIP_LIB:-plonky3-mobile_amd/libp3hip.so})"
  for rep in 1 2 3; do python3 tools/lde_unit_profile.py 20 1; done
  (cd /tmp && PYTHONPATH=$ROOT rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tw_$lib -o tw -- python3 $ROOT/tools/lde_unit_profile.py 20 1 > /dev/null 2>&1)
  grep "narrow_" $OUT/prof_tw_$lib/*/tw_kernel_stats.csv 2>/dev/null || grep -h "narrow_" $(find $OUT/prof_tw_$lib -name "*kernel_stats.csv") | cut -d, -f1-4
  SWEEP_W=2 SWEEP_LO=18 SWEEP_HI=22 python3 tools/lde_sweep.py 2>/dev/null | grep "blowup=2"
done
} > $OUT/r02_twiddle_ab.txt 2>&1
cat $OUT/r02_twiddle_ab.txt
