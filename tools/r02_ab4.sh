#!/bin/bash
out=gpurun_out/ab4; mkdir -p $out
export TMPDIR=/tmp
FOSPHOR_AMD_K1=6 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>$out/$label.err | python3 tools/bline.py $label; }
b k6_k23 FOSPHOR_AMD_K1=6
b k6_k23_again FOSPHOR_AMD_K1=6
b k1_k23 FOSPHOR_AMD_K1=1
b k5_k23 FOSPHOR_AMD_K1=5
b k6_nok23 FOSPHOR_AMD_K1=6 FOSPHOR_AMD_K23=0
b k6_k23_sub32 FOSPHOR_AMD_K1=6 FOSPHOR_AMD_SUB_LOG2=25
b k6_k23_sub128 FOSPHOR_AMD_K1=6 FOSPHOR_AMD_SUB_LOG2=27
