#!/bin/bash
# Run ON the GPU box: ablations of the 65536-point kernel (probe build from tools/ab_build.sh "probes:-DFOSPHOR_AMD_PROBES ...").
#   bash tools/r06_c5_probe.sh <lib name> <waves...>
lib=$1; shift
mkdir -p gpurun_out/ab
for w in "$@"; do
	for dbg in ${AB_DBG:-0 1 2 4 8 6 14 15}; do
		FOSPHOR_AMD_LIB=$PWD/build/ab/lib_$lib.so FOSPHOR_AMD_DBG_K1H=$dbg FOSPHOR_AMD_K1H_WAVES=$w python3 bench.py --config C5 --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/ab/p.json 2> gpurun_out/ab/p.err
		python3 tools/bline.py "C5 $lib w$w dbg $dbg" gpurun_out/ab/p.json
	done
done
