#!/bin/bash
out=gpurun_out/ab34; mkdir -p $out
b() { label=$1; cfg=$2; shift; shift; env "$@" timeout 200 python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b c3_$rep C3 X=1
b c3_ovl0_$rep C3 FOSPHOR_AMD_OVERLAP=0
b c3_noalt_$rep C3 FOSPHOR_AMD_ALT=0
b c3_ovl0_sub26_$rep C3 FOSPHOR_AMD_OVERLAP=0 FOSPHOR_AMD_SUB_LOG2=26
b c3_ovl0_sub28_$rep C3 FOSPHOR_AMD_OVERLAP=0 FOSPHOR_AMD_SUB_LOG2=28
done
