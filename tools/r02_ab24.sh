#!/bin/bash
out=gpurun_out/ab24; mkdir -p $out
b() { label=$1; cfg=$2; shift; shift; env "$@" timeout 200 python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2 3; do
b c2_new_$rep C2 X=1
b c2_old_$rep C2 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_prev.so
done
