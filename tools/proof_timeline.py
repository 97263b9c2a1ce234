#!/usr/bin/env python3
"""Timeline of the LAST proof in a rocprofv3 --kernel-trace CSV: start offset, gap to the previous kernel's end, duration, grid.
   python3 tools/proof_timeline.py <dir-or-csv> [first-kernel-substring=fib_trace_kernel] [min_us_to_print=0]
Kernels on other streams (e.g. the hiding prover's side-stream fills) appear interleaved by start time with their Stream_Id."""
import csv
import glob
import sys

src = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "fib_trace_kernel"
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
path = src if src.endswith(".csv") else glob.glob(src + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
# the proof starts a little before its first named kernel (seed / fill launches): take everything after the previous proof's last copy
start = idx[-1]
while start > 0 and int(rows[start]["Start_Timestamp"]) - int(rows[start - 1]["End_Timestamp"]) < 30000 and first not in rows[start - 1]["Kernel_Name"] \
        and "copyBuffer" not in rows[start - 1]["Kernel_Name"]:
    start -= 1
seq = rows[start:]
t0 = int(seq[0]["Start_Timestamp"])
prev_end = t0
busy = 0.0
for r in seq:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void p3::", "").replace("p3::", "").split("(")[0][:44]
    if (e - s) / 1e3 >= min_us:
        print("%9.1f us  gap %7.1f  dur %7.1f  grid %8s x %-4s stream %-3s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Grid_Size_X"],
              r["Workgroup_Size_X"], r.get("Stream_Id", "?"), name))
    busy += (e - s) / 1e3
    prev_end = max(prev_end, e)
print("launches %d, wall %.1f us, summed kernel time %.1f us" % (len(seq), (prev_end - t0) / 1e3, busy))
