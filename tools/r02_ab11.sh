#!/bin/bash
out=gpurun_out/ab11; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
b full X=1
b no_k3 FOSPHOR_AMD_DBG_SKIP=4
b no_k2 FOSPHOR_AMD_DBG_SKIP=8
b k1_only FOSPHOR_AMD_DBG_SKIP=2
b full_again X=1
