#!/usr/bin/env python3
"""Copy + kernel timeline of the LAST burst of activity in a rocprofv3 --kernel-trace --memory-copy-trace CSV pair
(usage: trace_timeline.py <dir>/<prefix>): events longer than 3 us, times in us from the burst's first event."""
import csv
import sys

pre = sys.argv[1]
ev = []
for r in csv.DictReader(open(pre + "_memory_copy_trace.csv")):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, e, r["Direction"].replace("MEMORY_COPY_", "")))
for r in csv.DictReader(open(pre + "_kernel_trace.csv")):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:28]))
ev.sort()
i0, end = 0, ev[0][1]
for i, (s, e, n) in enumerate(ev):
    if s - end > 1_000_000:
        i0 = i
    end = max(end, e)
t0 = ev[i0][0]
for s, e, n in ev[i0:]:
    if e - s > 3000:
        print("%9.1f .. %9.1f (%7.1f) %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
