#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root: the driver-style bench line, kernel-trace stats (+ union
# of overlapping dispatches) and the two HBM PMC passes for the current build.  Output under gpurun_out/$1/.
#   gpurun --timeout 1500 -- 'bash tools/profile_r02.sh r02a [--config C3]'
tag=${1:-r02x}; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 "$@" > "$out/bench.json" 2> "$out/bench.err"
python3 bench.py "$@" --no-cpu-baseline > "$out/bench_default.json" 2> "$out/bench_default.err"
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > "$out/bench_profiled.json" 2> "$out/kt.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/pmcF" -o p -- python3 bench.py --steps 4 --warmup 2 --precondition 0.05 --no-cpu-baseline --no-traffic-twin --no-extra-passes "$@" > /dev/null 2> "$out/pmcF.log"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$out/pmcW" -o p -- python3 bench.py --steps 4 --warmup 2 --precondition 0.05 --no-cpu-baseline --no-traffic-twin --no-extra-passes "$@" > /dev/null 2> "$out/pmcW.log"
find "$out/kt" -name "*_kernel_stats.csv" | head -1 | xargs cat > "$out/kernel_stats.csv"
python3 tools/kernel_union.py $(find "$out/kt" -name "*_kernel_trace.csv" | head -1) 3 > "$out/kernel_union.md"
python3 tools/pmc_summary.py $(find "$out/pmcF" "$out/pmcW" -name "*counter_collection.csv") > "$out/pmc.md"
# keep the merge-back small
find "$out" -name "*_kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*.db" -delete
python3 tools/bline.py driver_style < "$out/bench.json"; python3 tools/bline.py default < "$out/bench_default.json"; python3 tools/bline.py profiled < "$out/bench_profiled.json"
head -8 "$out/kernel_stats.csv"; cat "$out/kernel_union.md"; cat "$out/pmc.md"
