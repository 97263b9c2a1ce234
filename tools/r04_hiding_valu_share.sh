# Where a hiding proof's VALU instructions go: one prover (Keccak, 2^20-row trace) under rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES
# SQ_ACTIVE_INST_VALU (PMC only, its own run).  Usage: bash tools/r04_hiding_valu_share.sh [tag]
set -e
TAG=${1:-r04_hiding_valu}
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d $ROOT/gpurun_out/$TAG -- python3 $ROOT/tools/hiding_profile.py keccak 20 > $ROOT/gpurun_out/$TAG.log 2>&1
cd $ROOT
python3 - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
path = glob.glob("gpurun_out/%s/**/*counter_collection.csv" % tag, recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0].replace("void p3::", "").replace("p3::", "")
    if r["Counter_Name"] == "SQ_INSTS_VALU": acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_ACTIVE_INST_VALU": acc[k][2] += float(r["Counter_Value"])
n = 6
tot = sum(v[1] for v in acc.values()); tota = sum(v[2] for v in acc.values())
out = ["hiding prover, keccak, 2^20-row trace: VALU wave-instructions per proof %.1f M, SQ_ACTIVE_INST_VALU %.1f M cycles-units" % (tot / n / 1e6, tota / n / 1e6)]
for k, (c, v, a) in sorted(acc.items(), key=lambda kv: -kv[1][2])[:40]:
    out.append("%s %6.1f launches/proof %9.2f M wave-instr/proof %5.2f%%   active %9.2f M %5.2f%%" % (k[:50].ljust(52), c / n, v / n / 1e6, 100 * v / tot, a / n / 1e6, 100 * a / tota))
open("gpurun_out/%s.txt" % tag, "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
