#!/bin/bash
out=gpurun_out/ab15; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b base_$rep X=1
b noexact_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_noexact.so
b base_k1only_$rep FOSPHOR_AMD_DBG_SKIP=2
b noexact_k1only_$rep FOSPHOR_AMD_DBG_SKIP=2 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_noexact.so
done
