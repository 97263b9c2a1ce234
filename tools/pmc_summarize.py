#!/usr/bin/env python3
"""Turns the two rocprofv3 PMC passes of tools/pmc_probe.py (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs)
into the per-launch HBM-side traffic table that bench.py reports as roofline.traffic:
   python tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_lde_v2.json
Counter units are KiB; FETCH_SIZE is doubled (gfx950 reports half of the streamed read bytes,
MI355X_MICROARCH.md HBM section; the calibration kernels with known byte counts confirm it)."""
import csv
import glob
import json
import sys


def read(dirname, counter):
    path = glob.glob(dirname + "/**/*counter_collection.csv", recursive=True)[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


fetch, write = read(sys.argv[1], "FETCH_SIZE"), read(sys.argv[2], "WRITE_SIZE")
assert [k for _, k, _ in fetch] == [k for _, k, _ in write], "the two passes must run the same launches"


# FETCH_SIZE factor per kernel: 2 (aligned 128-byte requests tallied at 64 bytes) unless a calibration with a known byte count in
# the kernel's own access shape says otherwise.  tools/fetch_calib.hip (profiles/r04_fetch_calib.json): 4-byte lanes on 128-byte
# segments of rows that are NOT multiples of 128 bytes — K3 of the wide plan on a 2633-word-row matrix — are tallied in full
# (factor 1.016); the same access on 128-byte-multiple rows, and 16-byte lanes, are halved (factor 2.000).
def fetch_factor(kernel):
    return 1.0 if kernel.startswith("void p3::narrow_fwd2_kernel<8, 5, 1>") or kernel.startswith("void p3::narrow_fwd2_kernel<9, 5, 1>") else 2.0


launches = [{"kernel": k[:60], "fetch_bytes": fetch_factor(k) * f * 1024, "write_bytes": w * 1024, "fetch_factor": fetch_factor(k)}
            for (_, k, f), (_, _, w) in zip(fetch, write)]


def unit(names, which):
    """the `which`-th (from the end) consecutive run of launches whose kernel names start with `names` in order"""
    # integer and fp64 forms of the three launches count alike ("narrow_inv1_kernel<10" / "narrow64_inv1_kernel<10")
    idx = [i for i in range(len(launches) - len(names) + 1)
           if all(launches[i + j]["kernel"].replace("narrow64_", "narrow_").startswith(n) for j, n in enumerate(names))]
    i = idx[which]
    return launches[i:i + len(names)]


def first(prefix):
    return next(l for l in launches if prefix in l["kernel"])


out = {
    "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE runs of tools/pmc_probe.py; "
              "counters are KiB; FETCH_SIZE doubled (gfx950 reports 1/2 of streamed read bytes, MI355X_MICROARCH.md HBM "
              "section) EXCEPT for the wide plan's K3 on rows that are not multiples of 128 bytes, whose reads a calibration kernel of "
              "the same access shape shows tallied in full (profiles/r04_fetch_calib.json); summarised by tools/pmc_summarize.py",
    "calibration_bytes": {
        "fib_trace 2^24 rows (writes 128 MiB)": first("fib_trace_kernel"),
        "poseidon2 permute 2^22 states in place (reads 256 MiB, writes 256 MiB)": first("poseidon2_permute"),
        "leaf hash 2^24 x 2 (reads 128 MiB, writes 512 MiB)": first("leaf_hash"),
    },
}
k3 = ["void p3::narrow_inv1_kernel", "void p3::narrow_mid_kernel", "void p3::narrow_fwd2_kernel"]
for key, names, which, alg in (("cfg3_lde_2^24x2_blowup4", [n + "<12" for n in k3], -1, 4 * (1 << 24) * 2 * 5),
                               ("cfg2_lde_2^20x2_blowup2", [n + "<10" for n in k3], -1, 4 * (1 << 20) * 2 * 3)):
    u = unit(names, which)
    out[key] = {"launches": u, "total_bytes": sum(l["fetch_bytes"] + l["write_bytes"] for l in u), "algorithmic_bytes": alg}
# BASELINE configs[4]: every transform launch after the last 2^10-row fib_trace marker (tools/pmc_probe.py)
grids = {}
for d in (sys.argv[1],):
    import csv as _csv
    import glob as _glob
    for r in _csv.DictReader(open(_glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
        grids[int(r["Dispatch_Id"])] = int(r["Grid_Size"])
order = [i for i, _, _ in fetch]
marks = [k for k, (i, name, _) in enumerate(fetch) if "fib_trace_kernel" in name and grids.get(i, 0) <= 1024]
if marks:
    tail = [l for l in launches[marks[-1] + 1:] if "ntt_" in l["kernel"] or "narrow" in l["kernel"] or "bit_reverse" in l["kernel"]]
    out["cfg5_lde_2^16x2633_blowup2"] = {"launches": tail, "total_bytes": sum(l["fetch_bytes"] + l["write_bytes"] for l in tail),
                                         "algorithmic_bytes": 4 * (1 << 16) * 2633 * 3}
import hashlib  # noqa: E402
import os  # noqa: E402
_lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "plonky3-mobile_amd", "libp3hip.so")
out["lib_sha256"] = hashlib.sha256(open(_lib, "rb").read()).hexdigest()  # the build these counts belong to (bench_support._profile_json)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
out["src_sha256"] = __import__("build_id").build_id()["src_sha256"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: (v["total_bytes"], v["algorithmic_bytes"]) for k, v in out.items() if k.startswith("cfg")}))
