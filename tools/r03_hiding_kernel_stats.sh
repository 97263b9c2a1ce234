set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_j_hiding -o t -- python3 $ROOT/bench.py --hash keccak --hiding --no-cpu-baseline --steps 4 --warmup 1 > $ROOT/gpurun_out/r03_j_hiding.json 2> $ROOT/gpurun_out/r03_j_hiding.err
cd $ROOT
python3 - <<'PY'
import csv,json
d=json.loads(open("gpurun_out/r03_j_hiding.json").read().strip().splitlines()[-1]); print("proofs/s under profiler", d["value"])
rows=list(csv.DictReader(open("gpurun_out/r03_j_hiding/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:32]:
    print(r["Name"][:86].ljust(86), r["Calls"].rjust(6), "%9.1f us"%(float(r["AverageNs"])/1e3), "%5.1f%%"%(100*float(r["TotalDurationNs"])/tot))
PY
