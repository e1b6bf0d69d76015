#!/bin/bash
# Run ON the GPU box from the repo root: FETCH_SIZE calibration per load width, then kernel stats and the two PMC passes of
# `bench.py --config C3` for the build selected by the environment (FOSPHOR_AMD_K1W=0: the general kernel).
#   gpurun --timeout 900 -- 'bash tools/profile_c3.sh r03_c3'
tag=${1:-r03_c3}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
if [ -x tools/ubench/fetch_calib ] && [ ! -f "$out/calib.md" ]; then
	timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/calib" -o p -- ./tools/ubench/fetch_calib > /dev/null 2> "$out/calib.log"
	python3 tools/pmc_summary.py $(find "$out/calib" -name "*counter_collection.csv") > "$out/calib.md"
	rm -rf "$out/calib"
fi
timeout 200 python3 bench.py --config C3 --steps 20 --warmup 3 --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$out/kt" -o kt -- python3 bench.py --config C3 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-passes > "$out/bench_profiled.json" 2> "$out/kt.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/pmcF" -o p -- python3 bench.py --config C3 --steps 8 --warmup 2 --precondition 0.05 --no-cpu-baseline --no-extra-passes > /dev/null 2> "$out/pmcF.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace -f csv -d "$out/pmcW" -o p -- python3 bench.py --config C3 --steps 8 --warmup 2 --precondition 0.05 --no-cpu-baseline --no-extra-passes > /dev/null 2> "$out/pmcW.log"
find "$out" -name "*_kernel_stats.csv" | head -1 | xargs cat > "$out/kernel_stats.csv"
python3 tools/pmc_summary.py $(find "$out/pmcF" "$out/pmcW" -name "*counter_collection.csv") > "$out/pmc.md"
find "$out" -name "*_kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*.db" -delete
cat "$out/calib.md"; head -6 "$out/kernel_stats.csv"; cat "$out/pmc.md"
