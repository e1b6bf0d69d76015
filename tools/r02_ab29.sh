#!/bin/bash
out=gpurun_out/ab29; mkdir -p $out
b() { label=$1; cfg=$2; shift; shift; env "$@" timeout 200 python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
export FOSPHOR_AMD_K1H_FUSED=1
for rep in 1 2; do
b c5f_r8u2_$rep C5 X=1
b c5f_r4u8_$rep C5 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_r4u8.so
b c5f_r16u1_$rep C5 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_r16u1.so
b c5f_r8u4_$rep C5 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_r8u4.so
b c5two_r8u2_$rep C5 FOSPHOR_AMD_K1H_FUSED=0
b c5two_r16u1_$rep C5 FOSPHOR_AMD_K1H_FUSED=0 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_r16u1.so
done
