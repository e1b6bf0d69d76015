#!/bin/bash
out=gpurun_out/ab39; mkdir -p $out
b() { label=$1; shift; env "$@" timeout 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2 3; do
b dense_$rep X=1
b wavebits_$rep FOSPHOR_AMD_WAVEBITS=1
done
