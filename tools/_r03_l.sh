set -e
ROOT=$(pwd); export TMPDIR=/tmp
python -m pytest tests/test_gpu_ntt.py -x -q -k "narrow or large or headline or randomized_large" 2>&1 | tail -3
cd /tmp
for b in 0 1; do
  P3HIP_NTT_NARROW_BLOCKED12=$b rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_l_b$b -o t -- python3 $ROOT/tools/lde_probe.py 24:2:2 23:2:1 8 > $ROOT/gpurun_out/r03_l_b$b.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv
for b in (0,1):
    print("== BLOCKED12=%d"%b)
    rows=[r for r in csv.DictReader(open(f"gpurun_out/r03_l_b{b}/t_kernel_stats.csv")) if "narrow" in r["Name"]]
    for r in sorted(rows,key=lambda r:r["Name"]): print("  ",r["Name"][9:50].ljust(42), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3))
PY
