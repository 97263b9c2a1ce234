#!/usr/bin/env python3
"""Phase table from a NARROW_STAMPS dump (tools/_bin/libp3hip_stamps.so, P3HIP_NTT_STAMPS=<file>): median / p10 / p90
over workgroups of the s_memtime deltas between consecutive stamps of K1, and the spread of start / end times
(s_memrealtime, 100 MHz).   python tools/stamps_table.py <file>"""
import struct
import sys

import numpy as np

names = {0: "start", 1: "loads + twiddles arrived", 2: "two-level twiddles", 8: "round 1", 9: "exchange 1", 10: "round 2",
         11: "exchange 2", 3: "round 3", 4: "scale ladder", 5: "to_natural", 6: "stores acknowledged"}
order = [0, 1, 2, 8, 9, 10, 11, 3, 4, 5, 6]
data = open(sys.argv[1], "rb").read()
off, recs = 0, []
while off < len(data):
    n, w, added, tiles = struct.unpack_from("<4I", data, off)
    off += 16
    arr = np.frombuffer(data, dtype="<u8", count=tiles * 32, offset=off).reshape(tiles, 32)
    off += tiles * 256
    recs.append((n, w, added, tiles, arr))
n, w, added, tiles, arr = recs[-1]  # the last call (warm)
print("K1 of 2^%d x %d, blowup 2^%d: %d workgroups; s_memtime ticks" % (n, w, added, tiles))
prev = None
for i in order:
    if prev is not None:
        d = (arr[:, i] - arr[:, prev]).astype(np.int64)
        print("%-28s median %7d   p10 %7d   p90 %7d" % (names[i], np.median(d), np.percentile(d, 10), np.percentile(d, 90)))
    prev = i
tot = (arr[:, 6] - arr[:, 0]).astype(np.int64)
print("%-28s median %7d" % ("start -> stores acked", np.median(tot)))
rt0, rt1 = arr[:, 30].astype(np.int64), arr[:, 31].astype(np.int64)
print("realtime (10 ns ticks): first start %d, last start +%d, first end +%d, last end +%d; median lifetime %d" % (
    0, rt0.max() - rt0.min(), rt1.min() - rt0.min(), rt1.max() - rt0.min(), np.median(rt1 - rt0)))
