#!/bin/bash
out=gpurun_out/ab16; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b base_$rep X=1
b sets4_$rep FOSPHOR_AMD_SETS=4
b sets5_$rep FOSPHOR_AMD_SETS=5
b pipe3_$rep FOSPHOR_AMD_PIPE3=1
b pipe3_sets5_$rep FOSPHOR_AMD_PIPE3=1 FOSPHOR_AMD_SETS=5
b nowait_$rep FOSPHOR_AMD_DBG_NOWAIT=1
done
