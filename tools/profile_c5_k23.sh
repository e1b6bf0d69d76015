#!/bin/bash
# GPU box: C5 count / merge kernels with nothing beside them (no FFT kernels)
tag=${1:-c5_k23}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
export FOSPHOR_AMD_DBG_SKIP=1
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o kt -- python3 bench.py --config ${2:-C5} --steps 20 --warmup 5 --no-cpu-baseline --no-extra-passes --no-traffic-twin > "$out/bench_profiled.json" 2> "$out/kt.log"
find "$out/kt" -name "*_kernel_stats.csv" | head -1 | xargs cat > "$out/kernel_stats.csv"
find "$out" -name "*_kernel_trace.csv" -delete; find "$out" -name "*.db" -delete
python3 tools/bline.py ${tag}_profiled < "$out/bench_profiled.json"
grep fosphor "$out/kernel_stats.csv" | head -6
