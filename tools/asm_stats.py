#!/usr/bin/env python3
"""Instruction mix of one kernel from `hipcc -S` output, per basic block (label to label).

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -x hip --cuda-device-only -S -o build/asm/kernels.s gr-fosphor_amd/csrc/fosphor_kernels.hip
    python3 tools/asm_stats.py build/asm/kernels.s k1w_fft_binILb0 [min_instructions_per_block]

Columns: packed fp32 VALU (v_pk_*), other VALU, transcendental (v_log / v_exp / v_rcp ...), DS, vector memory, SALU, waits/barriers/nops.
The big blocks are the loop bodies; what the hot loop costs per thread and spectrum is read off them."""
import re
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith(("v_log", "v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_sleep")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*%s\S*:" % re.escape(sym), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    keys = ["pk", "valu", "trans", "ds", "vmem", "salu", "wait", "other"]
    blocks, cur, name = [], dict.fromkeys(keys, 0), lines[start].rstrip(":")
    total = dict.fromkeys(keys, 0)
    ops = {}
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        m = re.match(r"^(\.?[A-Za-z_][\w.$]*):", s)
        if m:
            blocks.append((name, cur))
            cur, name = dict.fromkeys(keys, 0), m.group(1)
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        c = classify(op)
        cur[c] += 1
        total[c] += 1
        ops.setdefault(name, {}).setdefault(op, 0)
        ops[name][op] += 1
    blocks.append((name, cur))
    print("%-28s" % "block" + "".join("%7s" % k for k in keys) + "   total")
    for n, c in blocks:
        t = sum(c.values())
        if t >= min_n:
            print("%-28s" % n[-28:] + "".join("%7d" % c[k] for k in keys) + "%8d" % t)
    print("%-28s" % "whole kernel" + "".join("%7d" % total[k] for k in keys) + "%8d" % sum(total.values()))
    if len(sys.argv) > 4:
        for o, n in sorted(ops.get(sys.argv[4], {}).items(), key=lambda x: -x[1]):
            print("   %-28s %d" % (o, n))


if __name__ == "__main__":
    main()
