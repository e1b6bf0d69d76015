#!/bin/bash
out=gpurun_out/ab7; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
b full X=1
b k2k3_nobytes FOSPHOR_AMD_DBG_SAME=1
b k1_only FOSPHOR_AMD_DBG_SKIP=2
b full_again X=1
