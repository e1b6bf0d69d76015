#!/bin/bash
out=gpurun_out/ab8; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
b v1_full X=1
b v2_full FOSPHOR_AMD_K1=2
b v2_full_tile32 FOSPHOR_AMD_K1=2 FOSPHOR_AMD_TILE=32
b v2_full_tile16 FOSPHOR_AMD_K1=2 FOSPHOR_AMD_TILE=16
b v2_only FOSPHOR_AMD_K1=2 FOSPHOR_AMD_DBG_SKIP=2
b v2_only_tile16 FOSPHOR_AMD_K1=2 FOSPHOR_AMD_DBG_SKIP=2 FOSPHOR_AMD_TILE=16
b v1_only X=1 FOSPHOR_AMD_DBG_SKIP=2
