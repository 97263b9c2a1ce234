#!/usr/bin/env python3
"""rocprofv3 PMC pass of tools/pmc_keccak_probe.py -> profiles/r02_pmc_keccak.json (formulas as in
tools/pmc_poseidon2_summarize.py):  python tools/pmc_keccak_summarize.py gpurun_out/pmc_kk profiles/r02_pmc_keccak.json"""
import csv
import glob
import json
import sys

CUS, XCDS = 256, 8
path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
disp = {}
for r in csv.DictReader(open(path)):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "grid": int(r["Grid_Size"]), "vgpr": int(r["VGPR_Count"]),
                                                "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
    d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
last = {}
for k in sorted(disp):
    d = disp[k]
    if "keccak" in d["kernel"]:
        last[(d["kernel"], d["grid"])] = d
out = []
for (name, grid), d in last.items():
    c = d["c"]
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    e = {"kernel": name.split("(")[0], "grid_threads": grid, "vgprs": d["vgpr"], "duration_us": d["ns"] / 1e3}
    if c.get("SQ_WAVES"):
        e["valu_insts_per_wave"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVES"]
    if gui:
        e["valu_busy_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / (CUS * gui)
        e["eff_clock_ghz"] = gui / d["ns"]
    if grid >= (1 << 20) and "coop" not in name:
        e["gperm_s"] = grid / d["ns"]
    out.append(e)
out.sort(key=lambda e: (e["kernel"], -e["grid_threads"]))
json.dump({"method": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES "
                     "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/pmc_keccak_probe.py; the LAST launch of each "
                     "(kernel, grid); valu_insts_per_wave = one permutation per lane for the leaf (rows of <= 34 elements) and "
                     "compress kernels", "kernels": out}, open(sys.argv[2], "w"), indent=1)
for e in out:
    print(e["kernel"][-34:], e["grid_threads"], "%.1f us" % e["duration_us"],
          {k: round(e[k], 3) for k in ("valu_insts_per_wave", "valu_busy_frac", "eff_clock_ghz", "gperm_s") if k in e})
