set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_t_keccak -o t -- python3 $ROOT/bench.py --hash keccak --no-cpu-baseline --steps 8 --warmup 1 > $ROOT/gpurun_out/r03_t_keccak.json 2> $ROOT/gpurun_out/r03_t_keccak.err
cd $ROOT
python3 - <<'PY'
import csv,json
d=json.loads(open("gpurun_out/r03_t_keccak.json").read().strip().splitlines()[-1]); print("proofs/s under profiler", d["value"])
rows=list(csv.DictReader(open("gpurun_out/r03_t_keccak/t_kernel_stats.csv")))
n=[int(r["Calls"]) for r in rows if "fib_quotient_kernel" in r["Name"]][0]
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("summed per proof %.2f ms"%(tot/n/1e6))
for r in rows[:24]:
    print(r["Name"].split("(")[0].replace("void p3::","").replace("p3::","")[:50].ljust(52), "%6.1f/proof"%(int(r["Calls"])/n), "%8.1f us"%(float(r["AverageNs"])/1e3), "%5.1f%%"%(100*float(r["TotalDurationNs"])/tot))
PY
