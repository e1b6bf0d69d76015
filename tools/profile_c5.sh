#!/bin/bash
# Run ON the GPU box from the repo root: bench line, kernel stats and the two PMC passes of `bench.py --config C5` for the
# library in $FOSPHOR_AMD_LIB (default: the in-tree build).  Prints a compact summary.
#   gpurun --timeout 900 -- 'bash tools/profile_c5.sh r03_c5'
tag=${1:-r03_c5}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
A="--config C5 --no-cpu-baseline"
timeout 200 python3 bench.py $A --steps 40 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o kt -- python3 bench.py $A --steps 40 --warmup 5 --no-extra-passes > "$out/bench_profiled.json" 2> "$out/kt.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/pmcF" -o p -- python3 bench.py $A --steps 8 --warmup 2 --precondition 0.05 --no-extra-passes > /dev/null 2> "$out/pmcF.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$out/pmcW" -o p -- python3 bench.py $A --steps 8 --warmup 2 --precondition 0.05 --no-extra-passes > /dev/null 2> "$out/pmcW.log"
find "$out" -name "*_kernel_stats.csv" | head -1 | xargs cat | cut -c1-160 > "$out/kernel_stats.csv"
python3 tools/pmc_summary.py $(find "$out/pmcF" "$out/pmcW" -name "*counter_collection.csv") | cut -c1-200 > "$out/pmc.md"
find "$out" -name "*_kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*.db" -delete
python3 tools/bline.py $tag < "$out/bench.json"; grep fosphor "$out/kernel_stats.csv" | head -6; grep -i "fosphor\|kernel" "$out/pmc.md"
