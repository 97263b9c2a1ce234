set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_v_tiny -o t -- python3 $ROOT/tools/hiding_bench.py 3 > $ROOT/gpurun_out/r03_v_tiny.txt 2>&1
cd $ROOT
grep hiding gpurun_out/r03_v_tiny.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r03_v_tiny/t_kernel_stats.csv")))
n=12  # 2 hashes x (1 + 5) proofs
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("launches per proof %.0f, summed kernel time per proof %.0f us"%(calls/n, tot/n/1e3))
for r in rows[:40]:
    print(r["Name"].split("(")[0].replace("void p3::","").replace("p3::","")[:52].ljust(54), "%5.1f/proof"%(int(r["Calls"])/n), "%7.1f us"%(float(r["AverageNs"])/1e3))
PY
