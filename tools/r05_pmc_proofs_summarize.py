#!/usr/bin/env python3
"""gpurun_out/r05_pmc_proof_<key>/**/counter_collection.csv -> gpurun_out/r05_pmc_proofs.json (tools/r05_pmc_proofs.sh)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
SPECS = {"cfg2": ("poseidon2", 20, 0, 6, 1), "cfg2_keccak": ("keccak", 20, 0, 6, 1), "cfg2_keccak_hiding": ("keccak", 20, 1, 6, 1),
         "cfg3": ("poseidon2", 24, 0, 2, 2)}
lib = os.path.join(ROOT, "plonky3-mobile_amd", "libp3hip.so")
out = {"method": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_VALU -- python3 tools/prove_n.py <hash> <log_n> <hiding> "
                 "<proofs> throughput <log_blowup>: ONE prover, throughput profile, bench.py's FRI parameters; sums over every launch of the "
                 "process divided by the number of proofs (table builds of the first proof included: < 0.1 %)",
       "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
       "src_sha256": __import__("build_id").build_id()["src_sha256"],
       "peak_wave_instr_per_s": 36e12 / 64,
       "peak_source": "36 T lane-ops/s measured (profiles/r01_microbench2_valu_issue_rates.txt) = 562.5 G wave-instructions/s",
       "workloads": {}}
for key, (hash_, log_n, hid, n, blow) in SPECS.items():
    paths = glob.glob(os.path.join(ROOT, "gpurun_out", "r05_pmc_proof_%s" % key, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(int)
    for r in csv.DictReader(open(paths[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void p3::", "").replace("p3::", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            launches[k] += 1
    total = sum(v["SQ_INSTS_VALU"] for v in acc.values())
    kern = []
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
        if v["SQ_INSTS_VALU"] / total < 0.002:
            continue
        kern.append({"kernel": k, "launches_per_proof": round(launches[k] / n, 2), "valu_wave_instr_per_proof": v["SQ_INSTS_VALU"] / n,
                     "share": round(v["SQ_INSTS_VALU"] / total, 4),
                     "valu_per_wave": round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1) if v["SQ_WAVES"] else None})
    hashk = sum(v["SQ_INSTS_VALU"] for k, v in acc.items() if any(s in k for s in ("keccak", "leaf_hash", "compress_layer", "tree_levels", "leaf_coop")))
    out["workloads"][key] = {"hash": hash_, "log_n": log_n, "hiding": bool(hid), "log_blowup": blow, "proofs": n,
                             "valu_wave_instr_per_proof": total / n, "hash_kernels_share": round(hashk / total, 4),
                             "proofs_per_s_at_peak_issue": (36e12 / 64) / (total / n), "kernels": kern}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_pmc_proofs.json"), "w"), indent=1)
for key, w in out["workloads"].items():
    print("%-20s %8.1f M VALU wave-instr per proof, hash kernels %.1f %%, %.1f proofs/s at peak issue" % (
        key, w["valu_wave_instr_per_proof"] / 1e6, 100 * w["hash_kernels_share"], w["proofs_per_s_at_peak_issue"]))
print("lib", out["lib_sha256"][:16])
