#!/bin/bash
out=gpurun_out/ab10; mkdir -p $out
FOSPHOR_AMD_K1=7 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
b base X=1
b k7_tile32 FOSPHOR_AMD_K1=7 FOSPHOR_AMD_TILE=32
b k7_tile64 FOSPHOR_AMD_K1=7 FOSPHOR_AMD_TILE=64
b k7_tile16 FOSPHOR_AMD_K1=7 FOSPHOR_AMD_TILE=16
b k7_only_tile32 FOSPHOR_AMD_K1=7 FOSPHOR_AMD_TILE=32 FOSPHOR_AMD_DBG_SKIP=2
b k7_only_tile64 FOSPHOR_AMD_K1=7 FOSPHOR_AMD_TILE=64 FOSPHOR_AMD_DBG_SKIP=2
b k7_only_tile16 FOSPHOR_AMD_K1=7 FOSPHOR_AMD_TILE=16 FOSPHOR_AMD_DBG_SKIP=2
b base_only X=1 FOSPHOR_AMD_DBG_SKIP=2
