#!/bin/bash
out=gpurun_out/ab22; mkdir -p $out
b() { label=$1; shift; env "$@" timeout 200 python3 bench.py --config C5 --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b c5_new_$rep X=1
b c5_prev_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_fused0.so
b c5_new_k1only_$rep FOSPHOR_AMD_DBG_SKIP=2
b c5_prev_k1only_$rep FOSPHOR_AMD_DBG_SKIP=2 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_fused0.so
done
