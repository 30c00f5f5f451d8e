#!/usr/bin/env python3
"""Idle gaps of a rocprofv3 kernel trace (CSV): every gap of at least <min_us> between the end of all earlier kernels and
the start of the next one, with the kernels on either side, inside the LAST step of the run (from the last launch of
<anchor>, default k_slice_width's first launch after the last k_cg_direction of the previous step is too fragile: the
anchor is the last occurrence of the kernel named on the command line, counted back <back> occurrences).
    trace_gaps.py <kernel_trace.csv> <min_us> <anchor substring> <occurrence (negative = from the end)> [length ms]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
min_us = float(sys.argv[2])
anchor = sys.argv[3]
occ = int(sys.argv[4])
length_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 1e9
hits = [i for i, r in enumerate(rows) if anchor in r[2]]
if not hits:
    sys.exit("no kernel matching %r" % anchor)
i0 = hits[occ]
t0 = rows[i0][0]
short = lambda n: n.split("(")[0].replace("void ", "").replace("bk::", "")[:40]
prev_end, prev_name = rows[i0][1], rows[i0][2]
busy = 0
idle = 0
out = []
last = t0
for s, e, name in rows[i0 + 1:]:
    if (s - t0) / 1e6 > length_ms:
        break
    gap = s - prev_end
    if gap > 0:
        idle += gap
    if gap / 1e3 >= min_us:
        out.append((s - t0, gap, short(prev_name), short(name)))
    if e > prev_end:
        busy += e - max(s, prev_end)
        prev_end, prev_name = e, name
    last = e
for at, gap, a, b in out:
    print("%10.1f us  idle %8.1f us   after %-40s before %s" % (at / 1e3, gap / 1e3, a, b))
print("span %.1f ms: busy %.1f ms, idle %.1f ms; %d gaps >= %.0f us hold %.1f ms" %
      ((last - t0) / 1e6, busy / 1e6, idle / 1e6, len(out), min_us, sum(g for _, g, _, _ in out) / 1e6))
