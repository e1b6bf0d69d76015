#!/bin/bash
# GPU box: what rocprofv3's FETCH_SIZE reports for a 1 GiB streaming read at 4, 8 and 16 bytes per lane, plain and
# non-temporal (tools/ubench/fetch_calib.hip, built here with hipcc).  Output: gpurun_out/<tag>/calib.md
tag=${1:-calib}
out=$PWD/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
[ -x tools/ubench/fetch_calib ] || hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip 2>/dev/null
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d "$out/calib" -o p -- ./tools/ubench/fetch_calib > /dev/null 2> "$out/calib.log"
timeout 60 python3 tools/pmc_summary.py $(find "$out/calib" -name "*counter_collection.csv") | cut -c1-160 > "$out/calib.md"
rm -rf "$out/calib"
cat "$out/calib.md"
