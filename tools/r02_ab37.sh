#!/bin/bash
out=gpurun_out/ab37; mkdir -p $out
b() { label=$1; shift; env "$@" timeout 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b sets3_$rep X=1
b sets2_$rep FOSPHOR_AMD_SETS=2
b sets2_sub25_$rep FOSPHOR_AMD_SETS=2 FOSPHOR_AMD_SUB_LOG2=25
b sets3_sub25_$rep FOSPHOR_AMD_SUB_LOG2=25
done
