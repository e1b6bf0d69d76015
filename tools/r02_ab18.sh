#!/bin/bash
out=gpurun_out/ab18; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --config C5 --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b c5_base_$rep X=1
b c5_k1only_$rep FOSPHOR_AMD_DBG_SKIP=2
b c5_noalt_$rep FOSPHOR_AMD_ALT=0
b c5_k1only_noalt_$rep FOSPHOR_AMD_DBG_SKIP=2 FOSPHOR_AMD_ALT=0
b c5_nok1_$rep FOSPHOR_AMD_DBG_SKIP=1
done
