#!/bin/bash
out=gpurun_out/ab30; mkdir -p $out
b() { label=$1; shift; env "$@" timeout 200 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
export FOSPHOR_AMD_DBG_SKIP=2
b k1only_full X=1
for v in 0 1 2 4 8 15; do b k1only_noexact_epi$v FOSPHOR_AMD_LIB=$PWD/build/ab/lib_epi$v.so; done
b k1only_full_again X=1
