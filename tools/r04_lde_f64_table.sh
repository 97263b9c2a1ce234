# coset LDE, integer vs fp64 butterflies vs fp64 rounds with word hand-overs, 2^21..2^24 rows: tools/lde_sweep.py
# three times in fresh processes (the switches are read once).  Output: gpurun_out/r04_lde_f64_vs_int.txt
set -e
out=gpurun_out/r04_lde_f64_vs_int.txt
: > $out
for cfg in "0 0" "7 0" "7 1"; do
  set -- $cfg
  echo "== P3HIP_NTT_NARROW_F64=$1 P3HIP_NTT_NARROW_F64_XW=$2" >> $out
  P3HIP_NTT_NARROW_F64=$1 P3HIP_NTT_NARROW_F64_XW=$2 SWEEP_W=2,4 SWEEP_LO=21 SWEEP_HI=25 python3 tools/lde_sweep.py >> $out 2>&1
done
python3 - <<'PY'
import re, collections
rows = collections.OrderedDict(); cur = None
for l in open("gpurun_out/r04_lde_f64_vs_int.txt"):
    if l.startswith("=="): cur = l.strip("= \n"); continue
    m = re.match(r"w=(\d+) blowup=(\d+) 2\^(\d+): ([0-9.]+) us", l)
    if m: rows.setdefault((int(m[1]), int(m[2]), int(m[3])), {})[cur] = float(m[4])
cols = list(next(iter(rows.values())).keys())
with open("gpurun_out/r04_lde_f64_vs_int.txt", "a") as f:
    f.write("\nshape".ljust(22) + "".join(c.replace("P3HIP_NTT_NARROW_", "").rjust(22) for c in cols) + "\n")
    for (w, b, n), d in rows.items():
        f.write(("w=%d b=%d 2^%d" % (w, b, n)).ljust(21) + "".join(("%.1f" % d.get(c, float("nan"))).rjust(22) for c in cols) + "\n")
print(open("gpurun_out/r04_lde_f64_vs_int.txt").read()[-1500:])
PY
