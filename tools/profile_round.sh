#!/bin/bash
# Run ON the GPU box (through gpurun) from the repo root: bench line, kernel-trace stats and the
# two HBM PMC passes for the current build.  Output under gpurun_out/$1/.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r01e'
tag=${1:-rXX}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o kt -- python3 bench.py --no-cpu-baseline > "$out/bench_profiled.json" 2> "$out/kt.log"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/pmcF" -o p -- python3 bench.py --steps 256 --warmup 128 --no-cpu-baseline > /dev/null 2> "$out/pmcF.log"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$out/pmcW" -o p -- python3 bench.py --steps 256 --warmup 128 --no-cpu-baseline > /dev/null 2> "$out/pmcW.log"
find "$out" -name "*_kernel_stats.csv" | head -1 | xargs cat > "$out/kernel_stats.csv"
python3 tools/pmc_summary.py $(find "$out/pmcF" "$out/pmcW" -name "*counter_collection.csv") > "$out/pmc.md"
# keep the merge-back small
find "$out" -name "*_kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*.db" -delete
cat "$out/bench.json"; cat "$out/kernel_stats.csv" | head -6; cat "$out/pmc.md"
