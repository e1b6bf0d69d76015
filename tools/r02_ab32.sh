#!/bin/bash
out=gpurun_out/ab32; mkdir -p $out
b() { label=$1; shift; env "$@" timeout 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b base_$rep X=1
b nopf_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_nopf.so
b k23_$rep FOSPHOR_AMD_K23=1
b nopf_k23_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_nopf.so FOSPHOR_AMD_K23=1
b nopf_k23_s17_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_nopf.so FOSPHOR_AMD_K23=1 FOSPHOR_AMD_SETS=4
done
