#!/usr/bin/env python3
"""Timeline of the LAST proof in a rocprofv3 --kernel-trace of tools/single_proof_latency.py: per kernel name the launches, the summed
duration and the summed idle time BEFORE each launch (start minus the latest end so far: with side streams kernels overlap, so idle
time counts only stretches where nothing ran), then the totals.
   rocprofv3 --kernel-trace -d gpurun_out/lat_tl -o t --output-format csv -- python3 tools/single_proof_latency.py 20 poseidon2 0 4 latency
   python3 tools/r05_latency_timeline.py gpurun_out/lat_tl <launches per proof or 0 = split at the largest gaps>"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# proofs are separated by host-side gaps (serialisation, Python): split at gaps > 100 us and take the last complete group
groups, cur, last_end = [], [], None
for s, e, n in rows:
    if last_end is not None and s - last_end > 100_000 and cur:
        groups.append(cur)
        cur = []
    cur.append((s, e, n))
    last_end = e if last_end is None else max(last_end, e)
if cur:
    groups.append(cur)
g = groups[-1]
t0, busy_end = g[0][0], g[0][0]
per = collections.OrderedDict()
idle_total = 0
for s, e, n in g:
    n = n.split("(")[0]
    idle = max(0, s - busy_end)
    idle_total += idle
    busy_end = max(busy_end, e)
    a = per.setdefault(n, [0, 0, 0])
    a[0] += 1; a[1] += e - s; a[2] += idle
span = busy_end - t0
print("last proof: %d launches, first start to last end %.1f us, chip idle between kernels %.1f us (%.1f %%)" % (len(g), span / 1e3, idle_total / 1e3, 100.0 * idle_total / span))
print("%-60s %8s %12s %14s" % ("kernel", "launches", "duration us", "idle before us"))
for n, (c, du, idl) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %8d %12.1f %14.1f" % (n[:60], c, du / 1e3, idl / 1e3))
