#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by tools/profile_round.sh) -> profiles/<tag>.md: bench lines, kernel stats of our kernels,
union table, PMC table.   python3 tools/make_profile_md.py <tag> "<title>" ["note"]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, title = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
out = os.path.join(ROOT, "gpurun_out", tag)


def line(name):
    p = os.path.join(out, name)
    if os.path.exists(p):
        for l in open(p):
            if l.startswith("{"):
                return l.strip()
    return None


b, bd, bp = line("bench.json"), line("bench_default.json"), line("bench_profiled.json")
rows = list(csv.reader(open(os.path.join(out, "kernel_stats.csv"))))
hdr, rows = rows[0], [r for r in rows[1:] if r and "fosphor_amd" in r[0] or (r and r[0].startswith("__amd"))]
keep = lambda txt: "\n".join(l for l in txt.split("\n") if "at::native" not in l)
union = keep(open(os.path.join(out, "kernel_union.md")).read())
pmc = keep(open(os.path.join(out, "pmc.md")).read())
md = "# %s\n\nProduced by `bash tools/profile_round.sh %s ...` on a 1xMI355X box (one gpurun call).  %s\n\n" % (title, tag, note)
md += "## 1. bench line, driver arguments (`--steps 20 --warmup 5`), un-profiled\n\n```\n%s\n```\n\n" % b
if bd:
    md += "Default arguments, same box: value %.0f MS/s.  " % json.loads(bd)["value"]
if bp:
    md += "Under `rocprofv3 --kernel-trace --stats`: value %.0f MS/s.\n\n" % json.loads(bp)["value"]
md += "## 2. rocprofv3 --kernel-trace --stats (whole process: pre-conditioning, warm-up, timed region, extra passes)\n\n```\n"
md += ",".join(hdr) + "\n" + "\n".join(",".join(r) for r in rows) + "\n```\n\n"
md += "Union of overlapping dispatches (`tools/kernel_union.py`, dispatches of 3 us and more):\n\n" + union + "\n"
md += "## 3. HBM counters (`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE GRBM_GUI_ACTIVE` passes; KiB per dispatch, means; "
md += "FETCH_SIZE x 2 = bytes on gfx950 for 4-, 8- and 16-byte-per-lane loads, plain and non-temporal: `profiles/r03_calib.md`)\n\n" + pmc
open(os.path.join(ROOT, "profiles", tag + ".md"), "w").write(md)
print("wrote profiles/%s.md (%d bytes)" % (tag, len(md)))
