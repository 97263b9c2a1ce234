#!/usr/bin/env python3
"""Per-kernel instruction histogram of the gfx950 code object of one .hip source (hipcc -S, device side only):
   python tools/isa_histogram.py plonky3-mobile_amd/csrc/mmcs.hip leaf_hash_f64 compress_layer_f64 poseidon2_permute_f64 \
       > profiles/r02_isa_histogram_poseidon2.json
For every kernel whose (demangled) name contains one of the given substrings: VGPR/SGPR counts, the static count per
mnemonic and per class, and the basic blocks in program order with their instruction counts and the label a trailing
branch jumps back to — rolled loops show up as a block (or run of blocks) ending in a backward s_cbranch, so the
DYNAMIC count of a permutation is sum(block count x trip count) with the trip counts of the source (4 + 3x4+1 + 4
rounds).  The measured dynamic count (SQ_INSTS_VALU / SQ_WAVES) is in profiles/r02_pmc_poseidon2.json."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile


def classify(m):
    if m.startswith("v_"):
        if "f64" in m:
            return "valu_f64"
        if m.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")):
            return "valu_int_mul"
        if m.startswith(("v_cvt", "v_rndne", "v_fract")):
            return "valu_cvt"
        if "dpp" in m or m.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")):
            return "valu_xlane"
        return "valu_other"
    if m.startswith("s_load") or m.startswith("s_buffer_load"):
        return "smem"
    if m.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if m.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep")):
        return "sync"
    if m.startswith("s_"):
        return "salu"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if m.startswith("ds_"):
        return "lds"
    return "other"


def main():
    src = sys.argv[1]
    wanted = sys.argv[2:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
                               "-S", "-o", out, src], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    meta = {}
    name = None
    for ln in text:
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1)
            meta[name] = {}
        for key in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size"):
            m = re.match(r"\s+\.%s:\s+(\d+)" % key, ln)
            if m and name:
                meta[name][key] = int(m.group(1))
    kernels = {}
    cur, blocks, label, order = None, None, None, None
    for ln in text:
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
        if m and m.group(1) in meta:
            cur = m.group(1)
            blocks = collections.OrderedDict()
            label = "entry"
            blocks[label] = []
            kernels[cur] = blocks
            continue
        if cur is None:
            continue
        if re.match(r"^\.Lfunc_end", ln):
            cur = None
            continue
        m = re.match(r"^(\.LBB\w+):", ln)
        if m:
            label = m.group(1)
            blocks[label] = []
            continue
        m = re.match(r"^\s+([a-z]\w+)(\s|$)", ln)
        if m and not ln.strip().startswith("."):
            blocks[label].append((m.group(1), ln.strip()))
    demangle = {}
    try:
        names = list(kernels)
        res = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"] + names, capture_output=True, text=True).stdout.split("\n")
        demangle = dict(zip(names, res))
    except Exception:
        pass
    report = {"source": src, "compiler": "hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S", "kernels": {}}
    for k, blocks in kernels.items():
        dn = demangle.get(k, k)
        if wanted and not any(w in dn for w in wanted):
            continue
        hist, cls = collections.Counter(), collections.Counter()
        blist = []
        labels = list(blocks)
        for lab, ins in blocks.items():
            target = None
            for mnem, full in ins:
                hist[mnem] += 1
                cls[classify(mnem)] += 1
                if mnem.startswith(("s_cbranch", "s_branch")):
                    t = full.split()[-1]
                    if t in labels and labels.index(t) <= labels.index(lab):
                        target = t
            c = collections.Counter(classify(m) for m, _ in ins)
            blist.append({"label": lab, "instructions": len(ins), "valu": sum(v for kk, v in c.items() if kk.startswith("valu")),
                          "valu_f64": c["valu_f64"], "loops_back_to": target})
        report["kernels"][dn] = dict(meta.get(k, {}), static_total=sum(hist.values()),
                                     static_valu=sum(v for kk, v in cls.items() if kk.startswith("valu")),
                                     by_class=dict(cls), by_mnemonic=dict(hist.most_common()), blocks=blist)
    json.dump(report, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
