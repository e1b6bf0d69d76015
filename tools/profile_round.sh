#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root: the driver-style bench line, kernel-trace stats (+ union of
# overlapping dispatches) and the two HBM PMC passes for the current build.  Output under gpurun_out/<tag>/, turned into
# profiles/<tag>.md by tools/make_profile_md.py.  Every step is bounded by `timeout`.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r03_c2; bash tools/profile_round.sh r03_c3 --config C3'
tag=${1:-rXX}; shift
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
T="timeout 300"
$T python3 bench.py --steps 20 --warmup 5 "$@" > "$out/bench.json" 2> "$out/bench.err"
$T python3 bench.py "$@" --no-cpu-baseline --no-other-configs > "$out/bench_default.json" 2> "$out/bench_default.err"
$T rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o kt -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs "$@" > "$out/bench_profiled.json" 2> "$out/kt.log"
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/pmcF" -o p -- python3 bench.py --steps 4 --warmup 2 --precondition 0.05 --no-cpu-baseline --no-traffic-twin --no-extra-passes --no-other-configs "$@" > /dev/null 2> "$out/pmcF.log"
$T rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$out/pmcW" -o p -- python3 bench.py --steps 4 --warmup 2 --precondition 0.05 --no-cpu-baseline --no-traffic-twin --no-extra-passes --no-other-configs "$@" > /dev/null 2> "$out/pmcW.log"
find "$out/kt" -name "*_kernel_stats.csv" | head -1 | xargs cat | cut -c1-200 > "$out/kernel_stats.csv"
timeout 120 python3 tools/kernel_union.py $(find "$out/kt" -name "*_kernel_trace.csv" | head -1) 3 > "$out/kernel_union.md"
timeout 120 python3 tools/pmc_summary.py $(find "$out/pmcF" "$out/pmcW" -name "*counter_collection.csv") | cut -c1-200 > "$out/pmc.md"
# keep the merge-back small
find "$out" -name "*_kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*.db" -delete
timeout 10 python3 tools/bline.py ${tag}_driver_style < "$out/bench.json"; timeout 10 python3 tools/bline.py ${tag}_default < "$out/bench_default.json"
grep fosphor "$out/kernel_stats.csv" | head -6 | cut -c1-150; grep -i "fosphor" "$out/pmc.md" | cut -c1-160
