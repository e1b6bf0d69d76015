#!/bin/bash
out=gpurun_out/ab19; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --config C5 --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b c5_new_$rep X=1
b c5_old_$rep FOSPHOR_AMD_LIB=$PWD/build/ab/lib_prev.so
b c5_g256t8_$rep FOSPHOR_AMD_K1H_GROUP=256 FOSPHOR_AMD_TILE=8
b c5_g128t8_$rep FOSPHOR_AMD_TILE=8
b c5_g256t4_$rep FOSPHOR_AMD_K1H_GROUP=256
b c5_nok1_$rep FOSPHOR_AMD_DBG_SKIP=1
done
