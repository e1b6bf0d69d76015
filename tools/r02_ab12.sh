#!/bin/bash
out=gpurun_out/ab12; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
b full X=1
b k2lds48 FOSPHOR_AMD_K2_LDS=48
b k2lds64 FOSPHOR_AMD_K2_LDS=64
b k2lds72 FOSPHOR_AMD_K2_LDS=72
b k2lds64_k3lds40 FOSPHOR_AMD_K2_LDS=64 FOSPHOR_AMD_K3_LDS=40
b k2lds64_k3lds60 FOSPHOR_AMD_K2_LDS=64 FOSPHOR_AMD_K3_LDS=60
b k3lds60 FOSPHOR_AMD_K3_LDS=60
b full_again X=1
