#!/usr/bin/env python3
"""Per-kernel busy time from a rocprofv3 --kernel-trace csv: calls, mean duration of a dispatch, and the UNION of
the dispatches' intervals divided by the calls.  K1s of consecutive sub-launches run on two streams and overlap,
so the mean duration counts shared time twice; union / calls is what a launch costs (bench.py computes the same
from hipEvents: roofline.k1_busy_ms_per_launch).

    python3 tools/kernel_union.py <..._kernel_trace.csv> [min_duration_us]
"""
import collections
import csv
import sys


def main(path, min_us=0.0):
    iv = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (e - s) / 1e3 >= min_us:
            iv[r["Kernel_Name"][:70]].append((s, e))
    print("| kernel | calls | mean duration us | union / calls us | overlap factor |")
    print("|---|---|---|---|---|")
    for name, v in sorted(iv.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        v.sort()
        tot = sum(e - s for s, e in v)
        union, cs, ce = 0, None, None
        for s, e in v:
            if ce is None or s > ce:
                if ce is not None:
                    union += ce - cs
                cs, ce = s, e
            elif e > ce:
                ce = e
        union += ce - cs
        print("| %s | %d | %.1f | %.1f | %.2f |" % (name, len(v), tot / len(v) / 1e3, union / len(v) / 1e3, tot / union))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
