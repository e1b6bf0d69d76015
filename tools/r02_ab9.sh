#!/bin/bash
out=gpurun_out/ab9; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>$out/$label.err | python3 tools/bline.py $label; }
b base X=1
b prio1 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_prio1.so
b prio3 FOSPHOR_AMD_LIB=$PWD/build/ab/lib_prio3.so
b base_again X=1
b prio3_again FOSPHOR_AMD_LIB=$PWD/build/ab/lib_prio3.so
