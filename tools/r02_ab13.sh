#!/bin/bash
out=gpurun_out/ab13; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
b base X=1
b ntst FOSPHOR_AMD_LIB=$PWD/build/ab/lib_ntst.so
b ntld FOSPHOR_AMD_LIB=$PWD/build/ab/lib_ntld.so
b ntboth FOSPHOR_AMD_LIB=$PWD/build/ab/lib_ntboth.so
b base_again X=1
