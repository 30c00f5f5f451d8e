#!/usr/bin/env python3
"""Prints a window of a rocprofv3 kernel trace (CSV): short kernel name, start offset, duration and the idle gap before
each kernel, plus the busy / idle totals of the window.   trace_window.py <kernel_trace.csv> <anchor kernel substring>
[occurrence index, default the middle one] [kernels to print, default 120]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
anchor = sys.argv[2]
hits = [i for i, r in enumerate(rows) if anchor in r[2]]
if not hits:
    sys.exit("no kernel matching %r" % anchor)
k = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) >= 0 else len(hits) // 2
cnt = int(sys.argv[4]) if len(sys.argv) > 4 else 120
if cnt == 0:     # intervals between successive anchor launches instead of a window
    prev = None
    for j, i in enumerate(hits):
        if prev is not None:
            print("%4d  +%9.1f us" % (j, (rows[i][0] - prev) / 1e3))
        prev = rows[i][0]
    print("total %.1f ms over %d anchors" % ((rows[hits[-1]][0] - rows[hits[0]][0]) / 1e6, len(hits)))
    sys.exit(0)
i0 = hits[min(k, len(hits) - 1)]
t0 = rows[i0][0]
busy = idle = 0
prev_end = rows[i0 - 1][1] if i0 else t0
for s, e, name in rows[i0:i0 + cnt]:
    short = name.split("(")[0].replace("void ", "").replace("bk::", "")
    gap = s - prev_end
    print("%9.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short))
    busy += e - s
    idle += max(0, gap)
    prev_end = max(prev_end, e)
print("window: busy %.1f us, idle %.1f us (%d kernels)" % (busy / 1e3, idle / 1e3, cnt))
