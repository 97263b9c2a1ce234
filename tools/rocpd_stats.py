#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, min, max in ns) of a rocprofv3 rocpd SQLite result file:
   python tools/rocpd_stats.py gpurun_out/prof/x_results.db [name-filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                  "group by %s order by 3 desc" % (name, name)).fetchall()
tot = sum(r[2] for r in rows) or 1
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
for r in rows:
    if flt in r[0]:
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (r[0][:110], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
