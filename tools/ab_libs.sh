#!/bin/bash
# Run ON the GPU box: interleaved A/B of prebuilt libraries (build/ab/lib_<name>.so; "cur" = the product library).
#   gpurun -- 'bash tools/ab_libs.sh C5 3 old cur'      config, repetitions, names...
cfg=$1; reps=$2; shift 2
mkdir -p gpurun_out/ab
for rep in $(seq 1 $reps); do
	for n in "$@"; do
		lib=$PWD/build/ab/lib_$n.so
		[ "$n" = cur ] && lib=$PWD/gr-fosphor_amd/libfosphor_amd.so
		FOSPHOR_AMD_LIB=$lib python3 bench.py --config $cfg --steps ${AB_STEPS:-200} --warmup 20 --no-cpu-baseline --no-other-configs ${AB_ARGS} > gpurun_out/ab/${cfg}_${n}_$rep.json 2> gpurun_out/ab/${cfg}_${n}_$rep.err
		python3 tools/bline.py "${cfg}_${n}_$rep" gpurun_out/ab/${cfg}_${n}_$rep.json
	done
done
