#!/usr/bin/env python3
"""Turns the rocprofv3 PMC pass of tools/pmc_poseidon2_probe.py (and of tools/_bin/clock_probe, the calibration) into
profiles/r02_pmc_poseidon2.json — the counter evidence behind bench.py's `valu_roofline`:
   python tools/pmc_poseidon2_summarize.py gpurun_out/pmc_p2 gpurun_out/pmc_p2_cal profiles/r02_pmc_poseidon2.json
Per kernel launch (the LAST launch of each (kernel, grid) pair):
   valu_insts_per_wave   = SQ_INSTS_VALU / SQ_WAVES            (dynamic VALU instructions of one wave = of one
                                                                permutation per lane for the compress / permute kernels)
   valu_busy_frac        = SQ_ACTIVE_INST_VALU / (CUs * GRBM_GUI_ACTIVE / XCDs)   [rocprofv3's own VALUBusy formula:
                           SQ_ACTIVE_INST_VALU is in quad-cycles summed over the 4 SIMDs of every CU, GRBM_GUI_ACTIVE
                           is reported as the sum over the 8 XCDs] — the fraction of elapsed SIMD time in which a VALU
                           instruction was executing; <= 1 by construction
   eff_clock_ghz         = GRBM_GUI_ACTIVE / XCDs / duration
   cycles_per_valu_inst  = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU  (issue cost of the average VALU instruction)
The calibration pass runs dependency-free v_fma_f64 / v_mul_lo_u32 chains with 8 waves per SIMD: its valu_busy_frac is
what "the VALU never idles" reads on these counters."""
import csv
import glob
import json
import sys

CUS, XCDS = 256, 8


def read(dirname):
    path = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)[0]
    disp = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            d = disp.setdefault(int(r["Dispatch_Id"]), {"kernel": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                                                        "wg": int(r["Workgroup_Size"]), "vgpr": int(r["VGPR_Count"]),
                                                        "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
            d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def derive(d):
    c = d["c"]
    out = {"kernel": d["kernel"][:90], "grid_threads": d["grid"], "workgroup": d["wg"], "vgprs": d["vgpr"], "duration_us": d["ns"] / 1e3,
           "counters": {k: c[k] for k in sorted(c)}}
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    if c.get("SQ_WAVES"):
        out["valu_insts_per_wave"] = c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVES"]
    if gui:
        out["valu_busy_frac"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / (CUS * gui)
        out["eff_clock_ghz"] = gui / d["ns"]
        if "SQ_BUSY_CU_CYCLES" in c:
            out["cu_busy_frac"] = c["SQ_BUSY_CU_CYCLES"] / (CUS * gui)
    if c.get("SQ_INSTS_VALU"):
        out["cycles_per_valu_inst"] = 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_INSTS_VALU"]
    return out


def last_by_shape(launches, needle):
    seen = {}
    for d in launches:
        if needle in d["kernel"]:
            seen[(d["kernel"], d["grid"])] = d
    return [derive(d) for d in seen.values()]


main_l = read(sys.argv[1])
cal_l = read(sys.argv[2])
rep = {"method": "rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES "
                 "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/pmc_poseidon2_probe.py (one pass; the "
                 "calibration is the same counter set over tools/_bin/clock_probe); summarised by tools/pmc_poseidon2_summarize.py",
       "calibration": [derive(d) for d in cal_l if "burn" in d["kernel"]],
       "permute_f64": last_by_shape(main_l, "poseidon2_permute_f64_kernel"),
       "leaf_hash_f64": last_by_shape(main_l, "leaf_hash_f64_kernel"),
       "compress_layer_f64": last_by_shape(main_l, "compress_layer_f64_kernel"),
       "tree_levels_coop": last_by_shape(main_l, "tree_levels_coop_kernel")}
big = [k for k in rep["leaf_hash_f64"] + rep["compress_layer_f64"] if k["grid_threads"] >= (1 << 20)]
busy = sum(k["counters"].get("SQ_ACTIVE_INST_VALU", 0.0) for k in big)
gui = sum(k["counters"].get("GRBM_GUI_ACTIVE", 0.0) / XCDS for k in big)
comp = [k for k in rep["compress_layer_f64"] if k["grid_threads"] >= (1 << 20)]
cal = [k.get("valu_busy_frac", 0.0) for k in rep["calibration"] if "burn<0>" in k["kernel"] or "ILi0" in k["kernel"]]
rep["summary"] = {
    "definition": "VALU-busy / elapsed over the leaf_hash_f64 and compress_layer_f64 launches of >= 2^20 lanes: "
                  "sum(SQ_ACTIVE_INST_VALU) / (256 CUs x sum(GRBM_GUI_ACTIVE / 8 XCDs))",
    "valu_busy_frac": busy / (CUS * gui) if gui else None,
    "valu_insts_per_permutation": (sum(k["valu_insts_per_wave"] for k in comp) / len(comp)) if comp else None,
    "calibration_valu_busy_frac_fma_f64": max(cal) if cal else None,
}
json.dump(rep, open(sys.argv[3], "w"), indent=1)
print(json.dumps(rep["summary"], indent=1))
for key in ("calibration", "permute_f64", "leaf_hash_f64", "compress_layer_f64"):
    for k in rep[key]:
        print(key, k["grid_threads"], "%.1f us" % k["duration_us"], {x: round(k[x], 3) for x in ("valu_insts_per_wave", "valu_busy_frac", "eff_clock_ghz", "cycles_per_valu_inst", "cu_busy_frac") if x in k})
