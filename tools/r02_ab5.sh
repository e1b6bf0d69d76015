#!/bin/bash
out=gpurun_out/ab5; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
b k1_only FOSPHOR_AMD_K1=1 FOSPHOR_AMD_DBG_SKIP=2
b k6_only FOSPHOR_AMD_K1=6 FOSPHOR_AMD_DBG_SKIP=2
b k5_only FOSPHOR_AMD_K1=5 FOSPHOR_AMD_DBG_SKIP=2
b k1_only_tile32 FOSPHOR_AMD_K1=1 FOSPHOR_AMD_DBG_SKIP=2 FOSPHOR_AMD_TILE=32
b k1_only_tile16 FOSPHOR_AMD_K1=1 FOSPHOR_AMD_DBG_SKIP=2 FOSPHOR_AMD_TILE=16
b k23_only FOSPHOR_AMD_K1=6 FOSPHOR_AMD_DBG_SKIP=1
b k2k3_only FOSPHOR_AMD_K1=6 FOSPHOR_AMD_DBG_SKIP=1 FOSPHOR_AMD_K23=0
b k1_full FOSPHOR_AMD_K1=1 FOSPHOR_AMD_K23=0
