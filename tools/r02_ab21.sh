#!/bin/bash
out=gpurun_out/ab21; mkdir -p $out
b() { label=$1; shift; env "$@" timeout 200 python3 bench.py --config C5 --steps 60 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
export FOSPHOR_AMD_DBG_SKIP=2
b k1only X=1
b nowait FOSPHOR_AMD_DBG_K1H=1
b noiq FOSPHOR_AMD_DBG_K1H=2
b nostore FOSPHOR_AMD_DBG_K1H=4
b noscratch FOSPHOR_AMD_DBG_K1H=8
b nowait_noscratch FOSPHOR_AMD_DBG_K1H=9
b nomem FOSPHOR_AMD_DBG_K1H=14
b nothing FOSPHOR_AMD_DBG_K1H=15
