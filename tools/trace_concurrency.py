#!/usr/bin/env python3
"""Concurrency of a rocprofv3 --kernel-trace CSV over the middle of the run: fraction of wall time with k kernels in flight,
with at least one chip-filling kernel (>= 4096 waves) in flight, and binned by work-items in flight.
   python3 tools/trace_concurrency.py <..._kernel_trace.csv> [lo_frac hi_frac]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lo_f, hi_f = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.4, 0.9)
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"] or 1) * int(r["Grid_Size_Z"] or 1)
    ev.append((s, 1, grid))
    ev.append((e, -1, grid))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
lo, hi = t0 + (t1 - t0) * lo_f, t0 + (t1 - t0) * hi_f
cur = curbig = curw = 0
last = None
hist, whist, bigtime = {}, {}, 0
for t, d, g in ev:
    if last is not None and t > lo and last < hi:
        a, b = max(last, lo), min(t, hi)
        if b > a:
            hist[cur] = hist.get(cur, 0) + (b - a)
            if curbig > 0:
                bigtime += b - a
            k = min(curw // 65536, 8)
            whist[k] = whist.get(k, 0) + (b - a)
    cur += d
    if g >= 64 * 4096:
        curbig += d
    curw += d * g
    last = t
tot = sum(hist.values())
print("window: %.0f%%..%.0f%% of the trace, %.1f ms" % (100 * lo_f, 100 * hi_f, tot / 1e6))
print("kernels in flight        :", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print("a >= 4096-wave kernel in flight: %.3f of the time" % (bigtime / tot))
print("work-items in flight / 65536 (8 = more): ", {k: round(v / tot, 3) for k, v in sorted(whist.items())})
