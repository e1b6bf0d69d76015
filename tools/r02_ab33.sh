#!/bin/bash
out=gpurun_out/ab33; mkdir -p $out
b() { label=$1; cfg=$2; shift; shift; env "$@" timeout 200 python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b c5_$rep C5 X=1
b c5_ovl0_$rep C5 FOSPHOR_AMD_OVERLAP=0
b c5_ovl0_strict_$rep C5 FOSPHOR_AMD_OVERLAP=0 BENCH_STRICT=1
done
