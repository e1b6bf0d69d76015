#!/bin/bash
out=gpurun_out/ab27; mkdir -p $out
b() { label=$1; cfg=$2; shift; shift; env "$@" timeout 200 python3 bench.py --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes 2>$out/$label.err | python3 tools/bline.py $label; }
for rep in 1 2; do
b c5_gateK3_b256_$rep C5 X=1
b c5_gateK2_b256_$rep C5 FOSPHOR_AMD_K1H_GATE_K2=1
b c5_gateK2_b512_$rep C5 FOSPHOR_AMD_K1H_GATE_K2=1 FOSPHOR_AMD_K3_BLOCKS=512
b c5_gateK2_b2048_$rep C5 FOSPHOR_AMD_K1H_GATE_K2=1 FOSPHOR_AMD_K3_BLOCKS=2048
b c5_gateK3_b2048_$rep C5 FOSPHOR_AMD_K3_BLOCKS=2048
b c5_two_$rep C5 FOSPHOR_AMD_K1H_FUSED=0
done
