#!/bin/bash
# GPU box: the NON-overlapped configuration (every K1 on one stream, 512 work-groups per launch = tiles of 32 spectra), where
# rocprofv3's plain AverageNs of K1 is the time a launch costs -- the cross-check for the union accounting of the default pipeline.
tag=${1:-r02_noalt}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
export FOSPHOR_AMD_ALT=0 FOSPHOR_AMD_TILE=32
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes --no-traffic-twin > "$out/bench_profiled.json" 2> "$out/kt.log"
find "$out/kt" -name "*_kernel_stats.csv" | head -1 | xargs cat > "$out/kernel_stats.csv"
python3 tools/kernel_union.py $(find "$out/kt" -name "*_kernel_trace.csv" | head -1) 3 > "$out/kernel_union.md"
find "$out" -name "*_kernel_trace.csv" -delete; find "$out" -name "*.db" -delete
python3 tools/bline.py noalt < "$out/bench.json"; python3 tools/bline.py noalt_profiled < "$out/bench_profiled.json"
grep fosphor "$out/kernel_stats.csv" | head -5; grep -v "at::native" "$out/kernel_union.md"
