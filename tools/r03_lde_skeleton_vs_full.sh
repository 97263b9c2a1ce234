set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
P3HIP_LIB=$ROOT/tools/_bin/libp3hip_skip.so P3HIP_NTT_NARROW_F64=0 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_d_skip -o t -- python3 $ROOT/tools/lde_probe.py 20:2:1 22:2:1 24:2:2 10 > $ROOT/gpurun_out/r03_d_skip.log 2>&1
P3HIP_NTT_NARROW_F64=0 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_d_full -o t -- python3 $ROOT/tools/lde_probe.py 20:2:1 22:2:1 24:2:2 10 > $ROOT/gpurun_out/r03_d_full.log 2>&1
cd $ROOT
for m in skip full; do echo == $m; grep narrow gpurun_out/r03_d_$m/t_kernel_stats.csv | cut -d, -f1-4; done
