#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (`*_results.db`) as a per-kernel stats table (markdown).

    python3 tools/rocpd_stats.py gpurun_out/prof_x/x_results.db > profiles/rNN_name_kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(top_kernels)")]
    rows = list(c.execute("select * from top_kernels"))
    print("| " + " | ".join(cols) + " |")
    print("|" + "---|" * len(cols))
    for r in rows:
        print("| " + " | ".join(("%.3f" % v) if isinstance(v, float) else str(v)[:90] for v in r) + " |")


if __name__ == "__main__":
    main(sys.argv[1])
