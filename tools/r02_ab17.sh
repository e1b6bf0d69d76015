#!/bin/bash
out=gpurun_out/ab17; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes $EXTRA 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
EXTRA="--mode frame"
b frame_$rep X=1
b frame_q8_$rep GPU_MAX_HW_QUEUES=8
b frame_force_$rep FOSPHOR_AMD_FORCE_EXCHANGE=1
b frame_force_q8_$rep FOSPHOR_AMD_FORCE_EXCHANGE=1 GPU_MAX_HW_QUEUES=8
EXTRA=""
b batch_$rep X=1
b batch_q8_$rep GPU_MAX_HW_QUEUES=8
done
