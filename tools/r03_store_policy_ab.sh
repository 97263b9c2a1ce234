set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
for v in plain wt1 wt2; do
  for m in 0 7; do
    lib=$ROOT/plonky3-mobile_amd/libp3hip.so; [ $v != plain ] && lib=$ROOT/tools/_bin/libp3hip_$v.so
    P3HIP_LIB=$lib P3HIP_NTT_NARROW_F64=$m rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_e_${v}_$m -o t -- python3 $ROOT/tools/lde_probe.py 20:2:1 20:4:1 24:2:2 10 > $ROOT/gpurun_out/r03_e_${v}_$m.log 2>&1
  done
done
cd $ROOT
python3 - <<'PY'
import csv
for v in ("plain","wt1","wt2"):
    for m in (0,7):
        print("==",v,"F64=%d"%m)
        rows=[r for r in csv.DictReader(open(f"gpurun_out/r03_e_{v}_{m}/t_kernel_stats.csv")) if "narrow" in r["Name"]]
        for r in sorted(rows,key=lambda r:r["Name"]): print("  ",r["Name"][9:50].ljust(42), r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3))
PY
