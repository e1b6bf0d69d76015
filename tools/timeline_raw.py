#!/usr/bin/env python3
"""Raw dispatch timeline (start, end, duration in us relative to the first listed dispatch) of the fosphor kernels in a
rocprofv3 --kernel-trace csv:  python3 tools/timeline_raw.py <kernel_trace.csv> [skip [count]]"""
import csv
import sys

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = []
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"]
    if "fosphor" not in n:
        continue
    short = n.split("fosphor_amd::")[-1].split("(")[0]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "")))
rows.sort()
rows = rows[skip:skip + count]
t0 = rows[0][0]
for s, e, n, q in rows:
    print("%9.1f %9.1f %8.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
