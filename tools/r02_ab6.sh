#!/bin/bash
out=gpurun_out/ab6; mkdir -p $out
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -2
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
b k1_only FOSPHOR_AMD_K1=1 FOSPHOR_AMD_DBG_SKIP=2
b k1_only_tile32 FOSPHOR_AMD_K1=1 FOSPHOR_AMD_DBG_SKIP=2 FOSPHOR_AMD_TILE=32
b k1_k2k3 FOSPHOR_AMD_K1=1 FOSPHOR_AMD_K23=0
b k1_k2k3_again FOSPHOR_AMD_K1=1 FOSPHOR_AMD_K23=0
b k1_k23 FOSPHOR_AMD_K1=1
b k1_k2k3_tile32 FOSPHOR_AMD_K1=1 FOSPHOR_AMD_K23=0 FOSPHOR_AMD_TILE=32
