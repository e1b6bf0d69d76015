#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv) as a markdown table.

    python3 tools/pmc_summary.py gpurun_out/pmc1/p1_counter_collection.csv [more.csv ...]
"""
import collections
import csv
import sys


def main(paths):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in paths:
        for r in csv.DictReader(open(path)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in agg.values() for c in k})
    print("| kernel | dispatches | " + " | ".join(counters) + " |")
    print("|---|---|" + "---|" * len(counters))
    for k, v in agg.items():
        n = max(len(x) for x in v.values())
        print("| %s | %d | " % (k, n) + " | ".join(("%.4g" % (sum(v[c]) / len(v[c]))) if c in v else "" for c in counters) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
