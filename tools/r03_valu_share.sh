set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $ROOT/gpurun_out/r03_u_valu -- python3 $ROOT/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --threads 1 --batch 8 > $ROOT/gpurun_out/r03_u_valu.json 2> $ROOT/gpurun_out/r03_u_valu.err
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
path = glob.glob("gpurun_out/r03_u_valu/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != "SQ_INSTS_VALU": continue
    k = r["Kernel_Name"].split("(")[0].replace("void p3::", "").replace("p3::", "")
    acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
n = acc["fib_quotient_kernel"][0]
tot = sum(v for _, v in acc.values())
print("proofs", n, "VALU wave-instructions per proof: %.1f M" % (tot / n / 1e6))
for k, (c, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print(k[:50].ljust(52), "%6.1f launches/proof" % (c / n), "%9.2f M wave-instr/proof" % (v / n / 1e6), "%5.2f%%" % (100 * v / tot))
PY
