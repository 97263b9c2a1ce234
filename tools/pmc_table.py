#!/usr/bin/env python3
"""Per-kernel table from a rocprofv3 --pmc pass (counter_collection.csv): mean counter values per launch, grouped by kernel
name, plus derived per-wave figures.   python tools/pmc_table.py <dir> [name filter]"""
import collections
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if flt not in k:
        continue
    k = re.sub(r"\(p3::\w+\)$", "", k).replace("void p3::", "")
    d = acc.setdefault(k, collections.defaultdict(list))
    d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    line = [k[:48].ljust(48), "n=%d" % len(next(iter(d.values())))]
    w = m.get("SQ_WAVES", 0) or 1
    for c, v in m.items():
        line.append("%s=%.4g" % (c.replace("SQ_", ""), v))
    if "SQ_INSTS_VALU" in m:
        line.append("| VALU/wave=%.0f" % (m["SQ_INSTS_VALU"] / w))
    if "SQ_INSTS_LDS" in m:
        line.append("LDS/wave=%.0f" % (m["SQ_INSTS_LDS"] / w))
    if "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        line.append("VALUbusy=%.3f" % (m["SQ_ACTIVE_INST_VALU"] / (256 * m["GRBM_GUI_ACTIVE"] / 8)))
    if "SQ_ACTIVE_INST_LDS" in m and "GRBM_GUI_ACTIVE" in m:
        line.append("LDSbusy=%.3f" % (m["SQ_ACTIVE_INST_LDS"] / (256 * m["GRBM_GUI_ACTIVE"] / 8)))
    print(" ".join(line))
