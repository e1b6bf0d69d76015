#!/bin/bash
# GPU box: raw kernel timeline of a short bench run.  profile_timeline.sh <tag> <skip> <count> [bench args...]
tag=$1; skip=$2; count=$3; shift; shift; shift
out=$PWD/gpurun_out/$tag; mkdir -p "$out"; export TMPDIR=/tmp
rocprofv3 --kernel-trace -f csv -d "$out/kt" -o kt -- python3 bench.py --steps 20 --warmup 5 --precondition 0.05 --no-cpu-baseline --no-extra-passes --no-traffic-twin "$@" > "$out/bench.json" 2> "$out/kt.log"
python3 tools/timeline_raw.py $(find "$out/kt" -name "*_kernel_trace.csv" | head -1) $skip $count | tee "$out/timeline.txt"
find "$out" -name "*_kernel_trace.csv" -delete; find "$out" -name "*.db" -delete
