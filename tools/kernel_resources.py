#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c x.hip -o /dev/null 2> remarks.txt; kernel_resources.py remarks.txt [filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip().rstrip("]").strip()
    if flt not in name:
        continue
    g = lambda k: re.search(k + r": (\d+)", b).group(1)
    try:
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dn = name
    dn = re.sub(r"\(p3::\w+\)$", "", dn).replace("void p3::", "")
    print(dn[:64].ljust(64), "VGPR", g("VGPRs").rjust(3), "AGPR", g("AGPRs").rjust(3), "spill", g("VGPRs Spill").rjust(3),
          "scratch", g(r"ScratchSize \[bytes/lane\]").rjust(4), "occ", g(r"Occupancy \[waves/SIMD\]"), "LDS", g(r"LDS Size \[bytes/block\]"))
