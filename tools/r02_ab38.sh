#!/bin/bash
out=gpurun_out/ab38; mkdir -p $out
b() { label=$1; shift; env "$@" timeout 200 python3 bench.py --mode frame --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b frame_g1_$rep X=1
b frame_g2_$rep FOSPHOR_AMD_FRAME_GROUP=2
b frame_g4_$rep FOSPHOR_AMD_FRAME_GROUP=4
b frame_g8_$rep FOSPHOR_AMD_FRAME_GROUP=8
done
python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>/dev/null | python3 tools/bline.py batch_mode
