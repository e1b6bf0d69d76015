#!/usr/bin/env python3
"""One compact line from a bench.py JSON line: bline.py <label> < file   or   bline.py <label> <file>   or   bline.py <file>
(a file name is never waited for on stdin: an earlier form of this tool did, and a gpurun call sat on it until its time limit)."""
import json
import os
import sys

label = sys.argv[1] if len(sys.argv) > 1 else ""
src = sys.stdin
if len(sys.argv) > 2:
    src = open(sys.argv[2])
elif label and os.path.isfile(label):
    src = open(label)
line = ""
for l in src:
    if l.startswith("{"):
        line = l
if not line:
    print("%-34s (no JSON line)" % label)
    sys.exit(0)
j = json.loads(line)
r = j["roofline"]
i = r.get("isolated") or {}
t = r.get("traffic_twin") or {}
print("%-34s %7.0f MS/s  whole %.3f | K1 busy %.1f us (%.3f) avg %.1f us ovl %.2f | K2 %.1f K3 %.1f | K1 alone %.1f twin %.1f | host %.2f" % (
    label, j["value"], r.get("whole_path_frac", 0), r["k1_busy_ms_per_launch"] * 1e3, r["frac"], r["k1_ms_per_launch"] * 1e3,
    r.get("k1_overlap") or 0, r["k2_ms_per_launch"] * 1e3, r["k3_ms_per_launch"] * 1e3,
    i.get("k1_ms_per_launch", 0) * 1e3, t.get("ms_per_launch", 0) * 1e3, j["config"]["host_submit_fraction"]))
