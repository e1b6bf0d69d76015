#!/bin/bash
out=gpurun_out/ab28; mkdir -p $out
b() { label=$1; shift; env "$@" timeout 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b base_$rep X=1
b w3_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_w3.so
b w3_s3_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_w3.so FOSPHOR_AMD_K1_STREAMS=3
b w3_t32_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_w3.so FOSPHOR_AMD_TILE=32
b w3_t32_noalt_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_w3.so FOSPHOR_AMD_TILE=32 FOSPHOR_AMD_ALT=0
b nopf_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_nopf.so
done
