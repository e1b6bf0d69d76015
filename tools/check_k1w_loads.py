#!/usr/bin/env python3
"""Build-time check of the 8192-point kernel's hand-issued IQ requests (gr-fosphor_amd/csrc/fosphor_kernels.hip, k1w_fft_bin).

The kernel requests the next spectrum's IQ with `buffer_load_dwordx2` written as inline assembly and waits for it with a hand-written
`s_waitcnt vmcnt(16)` at the top of the next iteration, so the compiler does not know that the destination registers are in flight (8-byte requests per row, or 16-byte requests per pair of rows).  That is
only correct if NOTHING reads or writes those registers between a request and the wait: a register-allocator copy in between would copy
stale data.  This script compiles the kernels to assembly and checks exactly that, for every instantiation:

    python3 tools/check_k1w_loads.py [kernels.s]        (without an argument: runs hipcc -S itself)

Exit status 0 and one line per kernel if the property holds."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gr-fosphor_amd", "csrc", "fosphor_kernels.hip")


def compile_asm(out):
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", "-o", out, SRC]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def regs(line):
    out = set()
    code = line.split(";")[0]
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", code):
        out.add(int(m.group(1)))
    return out


def check(path):
    src = open(path).read().splitlines()
    starts = [i for i, l in enumerate(src) if re.match(r"^_ZN11fosphor_amd11k1w_fft_binILi\d+EEEvNS_8K1ParamsE:", l)]
    if not starts:
        raise SystemExit("no k1w_fft_bin instantiation found in %s" % path)
    report = []
    for st in starts:
        end = next(i for i in range(st, len(src)) if src[i].startswith(".Lfunc_end"))
        body = src[st:end]
        name = re.match(r"^(\S+):", body[0]).group(1)
        waits = [i for i, l in enumerate(body) if "s_waitcnt vmcnt(16)" in l]
        if len(waits) != 1:
            raise SystemExit("%s: expected one counted wait, found %d" % (name, len(waits)))
        w = waits[0]
        # the loop: every label shortly above the wait that a branch BELOW the wait goes back to (the header, or a latch block falling into it)
        labels = {body[i].split(":")[0]: i for i in range(max(0, w - 200), w) if re.match(r"^\.LBB\d+_\d+:", body[i])}
        back = []
        for i in range(w, len(body)):
            m = re.match(r"^\s*s_c?branch\S*\s+(\.LBB\d+_\d+)\s*$", body[i].split(";")[0].rstrip())
            if m and m.group(1) in labels:
                back.append((i, labels[m.group(1)]))
        if not back:
            raise SystemExit("%s: no back edge above the wait" % name)
        tail = max(i for i, _ in back)
        header = min(h for _, h in back)
        label = body[header].split(":")[0]
        loads = [i for i in range(w, tail) if re.search(r"buffer_load_dwordx[24] ", body[i])]
        if not loads:
            raise SystemExit("%s: no IQ request inside the loop" % name)
        dests, bad = set(), []
        for ld in loads:		# a register is in flight from ITS request to the wait
            m = re.search(r"buffer_load_dwordx[24] v\[(\d+):(\d+)\]", body[ld])
            d = set(range(int(m.group(1)), int(m.group(2)) + 1))
            dests |= d
            for i in list(range(ld + 1, tail + 1)) + list(range(header, w)):
                l = body[i].strip()
                if not l or l.startswith((";", ".")):
                    continue
                if regs(l) & d:
                    bad.append((i, l))
        # behind the loop: the last iteration's requests are still in flight until the hand-written vmcnt(0)
        fin = next((i for i in range(tail, len(body) - 1) if "ASMSTART" in body[i] and "s_waitcnt vmcnt(0)" in body[i + 1]), None)
        if fin is None:
            raise SystemExit("%s: no hand-written vmcnt(0) behind the loop" % name)
        for i in range(tail + 1, fin):
            l = body[i].strip()
            if l and not l.startswith((";", ".")) and regs(l) & dests:
                bad.append((i, l))
        if bad:
            for i, l in bad[:10]:
                print("%s +%d: %s" % (name, i, l), file=sys.stderr)
            raise SystemExit("%s: %d instruction(s) touch an in-flight IQ register between its request and the wait" % (name, len(bad)))
        # The counted wait's immediate must equal what an ODD spectrum's epilogue issues BEHIND the requests: its index stores (inline
        # assembly `global_store_short`, one copy of the three pieces for the early and one for the late waves of a SIMD).  A change
        # that packs the stores must change the immediate with them (kK1wIdxStores), or the wait returns before the IQ has landed.
        imm = int(re.search(r"vmcnt\((\d+)\)", body[w]).group(1))
        idx_stores, in_asm = 0, False
        for i in range(header, tail + 1):
            l = body[i]
            if "ASMSTART" in l:
                in_asm = True
            elif "ASMEND" in l:
                in_asm = False
            elif in_asm and re.match(r"\s*global_store_short\s", l):
                idx_stores += 1
        if idx_stores != 2 * imm:
            raise SystemExit("%s: the counted wait is vmcnt(%d) but the loop holds %d hand-issued index stores (expected 2 x %d: early and late copy)"
                             % (name, imm, idx_stores, imm))
        # Spills: none inside the spectrum loop but for the fft_out test hook's block (the one with the global_store_dwordx2 of the
        # spectrum itself); the kernel sits at 256 registers, so a compiler bump could move one into the hot path unnoticed.
        blocks, cur = [], []
        for i in range(header, tail + 1):
            if re.match(r"^\.LBB\d+_\d+:", body[i]) and cur:
                blocks.append(cur)
                cur = []
            cur.append(body[i])
        blocks.append(cur)
        hot_spills = sum(1 for b in blocks if any("scratch_" in l for l in b) and not any("global_store_dwordx2" in l for l in b))
        if hot_spills:
            raise SystemExit("%s: %d basic block(s) of the spectrum loop spill or reload outside the fft_out test path" % (name, hot_spills))
        scratch = next((int(m.group(1)) for l in src[st:end + 200] for m in [re.search(r";\s*ScratchSize:\s*(\d+)", l)] if m), -1)
        if scratch > 64:
            raise SystemExit("%s: ScratchSize %d bytes per lane (budget: 64, all of it outside the spectrum loop)" % (name, scratch))
        report.append("%s: %d requests -> v%d..v%d, untouched until the wait (loop %s, %d lines); vmcnt(%d) = %d index stores per wave; "
                      "no spill in the loop's product path; ScratchSize %d"
                      % (name, len(loads), min(dests), max(dests), label, tail - header, imm, idx_stores // 2, scratch))
    return report


if __name__ == "__main__":
    if len(sys.argv) > 1:
        lines = check(sys.argv[1])
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "kernels.s")
            compile_asm(out)
            lines = check(out)
    print("\n".join(lines))
