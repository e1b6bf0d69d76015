#!/bin/bash
out=gpurun_out/ab14; mkdir -p $out
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b n3s2_$rep FOSPHOR_AMD_SETS=3 FOSPHOR_AMD_K1_STREAMS=2
b n3s3_$rep FOSPHOR_AMD_SETS=3 FOSPHOR_AMD_K1_STREAMS=3
b n4s3_$rep FOSPHOR_AMD_SETS=4 FOSPHOR_AMD_K1_STREAMS=3
b n5s3_$rep FOSPHOR_AMD_SETS=5 FOSPHOR_AMD_K1_STREAMS=3
b q8n3s2_$rep GPU_MAX_HW_QUEUES=8 FOSPHOR_AMD_SETS=3 FOSPHOR_AMD_K1_STREAMS=2
b q8n4s3_$rep GPU_MAX_HW_QUEUES=8 FOSPHOR_AMD_SETS=4 FOSPHOR_AMD_K1_STREAMS=3
b q8n5s3_$rep GPU_MAX_HW_QUEUES=8 FOSPHOR_AMD_SETS=5 FOSPHOR_AMD_K1_STREAMS=3
b q8n5s4_$rep GPU_MAX_HW_QUEUES=8 FOSPHOR_AMD_SETS=5 FOSPHOR_AMD_K1_STREAMS=4
done
