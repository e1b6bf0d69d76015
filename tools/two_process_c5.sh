#!/bin/bash
# GPU box: two PROCESSES running the fused 65536-point path on the same GPU at the same time (each bounded by timeout)
timeout 120 python3 bench.py --config C5 --steps 200 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes > gpurun_out/two_a.json 2> gpurun_out/two_a.err &
pa=$!
timeout 120 python3 bench.py --config C5 --steps 200 --warmup 10 --no-cpu-baseline --no-traffic-twin --no-extra-passes > gpurun_out/two_b.json 2> gpurun_out/two_b.err &
pb=$!
wait $pa; ra=$?; wait $pb; rb=$?
echo "exit codes: $ra $rb"
python3 tools/bline.py proc_a < gpurun_out/two_a.json; python3 tools/bline.py proc_b < gpurun_out/two_b.json
grep -h "fosphor_amd\]" gpurun_out/two_a.err gpurun_out/two_b.err | head -3
