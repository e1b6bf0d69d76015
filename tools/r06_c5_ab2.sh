#!/bin/bash
# Run ON the GPU box: C5, libraries x work-group forms interleaved.   bash tools/r06_c5_ab2.sh <reps> <lib names...>   ("cur" = product)
reps=$1; shift
mkdir -p gpurun_out/ab
for rep in $(seq 1 $reps); do
	for n in "$@"; do
		lib=$PWD/build/ab/lib_$n.so
		[ "$n" = cur ] && lib=$PWD/gr-fosphor_amd/libfosphor_amd.so
		for w in ${AB_WAVES:-8 4}; do
			FOSPHOR_AMD_LIB=$lib FOSPHOR_AMD_K1H_WAVES=$w python3 bench.py --config C5 --steps ${AB_STEPS:-200} --warmup 20 --no-cpu-baseline --no-other-configs ${AB_ARGS} > gpurun_out/ab/C5_${n}_w${w}_$rep.json 2> gpurun_out/ab/C5_${n}_w${w}_$rep.err
			python3 tools/bline.py "C5_${n}_w${w}_$rep" gpurun_out/ab/C5_${n}_w${w}_$rep.json
		done
	done
done
