#!/bin/bash
# Run ON the GPU box: C5 with the 65536-point kernel's two forms interleaved (FOSPHOR_AMD_K1H_FORM: 0 specialised waves, 1 one program).
#   bash tools/r06_c5_forms.sh <reps> [lib name]
reps=${1:-3}; lib=${2:-cur}
L=$PWD/build/ab/lib_$lib.so; [ "$lib" = cur ] && L=$PWD/gr-fosphor_amd/libfosphor_amd.so
mkdir -p gpurun_out/ab
for rep in $(seq 1 $reps); do
	for f in 1 0; do
		FOSPHOR_AMD_LIB=$L FOSPHOR_AMD_K1H_FORM=$f python3 bench.py --config C5 --steps ${AB_STEPS:-200} --warmup 20 --no-cpu-baseline --no-other-configs ${AB_ARGS} > gpurun_out/ab/C5_form${f}_$rep.json 2> gpurun_out/ab/C5_form${f}_$rep.err
		python3 tools/bline.py "C5_${lib}_form${f}_$rep" gpurun_out/ab/C5_form${f}_$rep.json
	done
done
