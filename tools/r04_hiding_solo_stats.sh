# One hiding prover (Keccak, 2^20-row trace) alone on the chip under rocprofv3 --kernel-trace --stats: per-proof kernel time by kernel,
# i.e. what each phase costs when nothing overlaps it.  Usage: bash tools/r04_hiding_solo_stats.sh [tag]
set -e
TAG=${1:-r04_hiding_solo}
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/$TAG -o t -- python3 $ROOT/tools/hiding_profile.py keccak 20 > $ROOT/gpurun_out/$TAG.log 2>&1
cd $ROOT
python3 - $TAG <<'PY'
import csv, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open("gpurun_out/%s/t_kernel_stats.csv" % tag)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
out = ["one hiding prover, keccak, 2^20-row trace, 6 proofs: %.3f ms of kernel time per proof" % (tot / 6e6)]
for r in rows[:48]:
    out.append("%s %6s calls %9.1f us avg %9.1f us/proof %5.1f%%" % (r["Name"][:84].ljust(84), r["Calls"], float(r["AverageNs"]) / 1e3,
               float(r["TotalDurationNs"]) / 6e3, 100 * float(r["TotalDurationNs"]) / tot))
open("gpurun_out/%s.txt" % tag, "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
