# VALU instructions of the coset-LDE unit at the three BASELINE shapes (cfg2 2^20x2 b2, cfg3 2^24x2 b4, cfg5 2^16x2633 b2):
# rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE (PMC only, one run per shape) over tools/lde_probe.py,
# summarised per kernel into gpurun_out/r05_pmc_lde_valu.json (copy to profiles/).  bench.py's roofline.valu_frac reads that file.
set -e
ROOT=$(pwd); export TMPDIR=/tmp
cd /tmp
for spec in cfg2:20:2:1 cfg3:24:2:2 cfg5:16:2633:1; do
  name=${spec%%:*}; shape=${spec#*:}
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/r05_lde_valu_$name -- python3 $ROOT/tools/lde_probe.py $shape 4 > $ROOT/gpurun_out/r05_lde_valu_$name.log 2>&1
done
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, json, sys
sys.path.insert(0, "tools")
from build_id import build_id
out = dict(build_id(), **{"method": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -- python3 tools/lde_probe.py <shape> 4, "
                 "one run per shape (tools/r05_lde_valu.sh); means per launch over the unit's kernels; wave-instructions = SQ_INSTS_VALU",
       "peak_wave_instr_per_s": 36e12 / 64})
shapes = {"cfg2": (20, 2, 1), "cfg3": (24, 2, 2), "cfg5": (16, 2633, 1)}
for name, (n, w, ab) in shapes.items():
    path = glob.glob("gpurun_out/r05_lde_valu_%s/**/*counter_collection.csv" % name, recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "narrow" not in k and "ntt_" not in k:
            continue
        k = k.split("(")[0].replace("void p3::", "")
        acc.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    h = 1 << n
    butterflies = (h // 2) * n * w + ((h << ab) // 2) * n * w  # what the two-digit plan executes: 2^ab size-h forward transforms (no stage over zero padding)
    kern, total = {}, 0.0
    for k, d in acc.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        kern[k] = {"valu_wave_instr_per_launch": m["SQ_INSTS_VALU"], "waves": m["SQ_WAVES"], "valu_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
                   "valu_busy": m["SQ_ACTIVE_INST_VALU"] / (256 * m["GRBM_GUI_ACTIVE"] / 8), "launches_seen": len(d["SQ_WAVES"])}
        total += m["SQ_INSTS_VALU"]
    out[name] = {"shape": "2^%d x %d, blowup %d" % (n, w, 1 << ab), "kernels": kern, "unit_valu_wave_instr": total, "butterflies": butterflies,
                 "valu_lane_instr_per_butterfly": total * 64 / butterflies,
                 "algorithmic_bytes": 4 * h * w * (1 + (1 << ab)),
                 "unit_us_at_peak_issue": total / (36e12 / 64) * 1e6}
    out[name]["hbm_frac_ceiling_at_this_instruction_count"] = out[name]["algorithmic_bytes"] / (out[name]["unit_us_at_peak_issue"] * 1e-6) / 8e12
json.dump(out, open("gpurun_out/r05_pmc_lde_valu.json", "w"), indent=1)
for name in shapes:
    o = out[name]
    print(name, o["shape"], "VALU wave-instr %.2f M, %.2f lane-instr per butterfly, %.1f us at peak issue, HBM-frac ceiling %.3f" %
          (o["unit_valu_wave_instr"] / 1e6, o["valu_lane_instr_per_butterfly"], o["unit_us_at_peak_issue"], o["hbm_frac_ceiling_at_this_instruction_count"]))
PY
