#!/usr/bin/env python3
"""Timeline of the K1/K2/K3 pipeline from a rocprofv3 --kernel-trace csv (`*_kernel_trace.csv`).

    python3 tools/timeline.py gpurun_out/x/kt_kernel_trace.csv [first_launch [count]]

For each K1 dispatch: duration, idle gap since the previous K1 ended, and when the K2 / K3 that
consume it started and ended relative to the K1's end.  Used to see where the two-stream pipeline
 leaves the stream of K1s waiting.
"""
import csv
import sys


def main(path, first=0, count=24):
    rows = []
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        k = "K1" if "k1" in name and "fft_bin" in name else "K2" if "k2_count" in name else \
            "K3" if "k3_merge" in name else "K2b" if "k2b" in name else None
        if k:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
    rows.sort()
    k1 = [r for r in rows if r[2] == "K1"]
    k2 = [r for r in rows if r[2] == "K2"]
    k3 = [r for r in rows if r[2] == "K3"]
    print("| # | K1 us | gap before K1 us | K2 start-after-K1-end us | K2 us | K3 start-after-K2-end us | K3 us | K3 end after K1 end us |")
    print("|---|---|---|---|---|---|---|---|")
    for i in range(first, min(len(k1), first + count)):
        s, e, _ = k1[i]
        gap = (s - k1[i - 1][1]) / 1e3 if i else 0.0
        a = k2[i] if i < len(k2) else None
        b = k3[i] if i < len(k3) else None
        print("| %d | %.1f | %.1f | %s | %s | %s | %s | %s |" % (
            i, (e - s) / 1e3, gap,
            "%.1f" % ((a[0] - e) / 1e3) if a else "", "%.1f" % ((a[1] - a[0]) / 1e3) if a else "",
            "%.1f" % ((b[0] - a[1]) / 1e3) if a and b else "", "%.1f" % ((b[1] - b[0]) / 1e3) if b else "",
            "%.1f" % ((b[1] - e) / 1e3) if b else ""))


if __name__ == "__main__":
    main(sys.argv[1], *[int(x) for x in sys.argv[2:4]])
