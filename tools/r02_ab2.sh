#!/bin/bash
out=gpurun_out/ab2; mkdir -p $out
FOSPHOR_AMD_K1=5 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>$out/$label.err | python3 tools/bline.py $label; }
b k1_default X=1
b k1d FOSPHOR_AMD_K1=5
b k1d_again FOSPHOR_AMD_K1=5
b k1d_tile32 FOSPHOR_AMD_K1=5 FOSPHOR_AMD_TILE=32
b k1d_alt0_tile16 FOSPHOR_AMD_K1=5 FOSPHOR_AMD_TILE=16 FOSPHOR_AMD_ALT=0
b k1d_tile32_alt0 FOSPHOR_AMD_K1=5 FOSPHOR_AMD_TILE=32 FOSPHOR_AMD_ALT=0
b k1d_sub128 FOSPHOR_AMD_K1=5 FOSPHOR_AMD_SUB_LOG2=27
b k1d_sub32 FOSPHOR_AMD_K1=5 FOSPHOR_AMD_SUB_LOG2=25
