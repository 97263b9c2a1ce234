# the reference's own instance (n = 8, x = 21: fib_air.rs:56-57) under the kernel trace: launches and timeline of one proof
set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r04_tiny -o t -- python3 $ROOT/tools/hiding_profile.py keccak 3 > $ROOT/gpurun_out/r04_tiny.log 2>&1
cd $ROOT
python3 tools/proof_timeline.py gpurun_out/r04_tiny > gpurun_out/r04_tiny_timeline.txt
tail -n 100 gpurun_out/r04_tiny_timeline.txt
