#!/bin/bash
out=gpurun_out/ab23; mkdir -p $out
b() { label=$1; cfg=$2; shift; shift; env "$@" timeout 200 python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin $EXTRA 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
EXTRA=""
b c2_mask_$rep C2 X=1
b c2_nomask_$rep C2 FOSPHOR_AMD_NO_ROWMASK=1
EXTRA="--no-extra-passes"
b c3_mask_$rep C3 X=1
b c3_nomask_$rep C3 FOSPHOR_AMD_NO_ROWMASK=1
b c5_mask_$rep C5 X=1
b c5_nomask_$rep C5 FOSPHOR_AMD_NO_ROWMASK=1
done
