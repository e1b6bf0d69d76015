#!/bin/bash
# Run ON the GPU box: C5 with work-groups of 4 and of 8 waves, interleaved (one library, FOSPHOR_AMD_K1H_WAVES).
#   gpurun -- 'bash tools/r06_c5_ab.sh 3'
reps=${1:-3}
mkdir -p gpurun_out/ab
for rep in $(seq 1 $reps); do
	for w in 8 4; do
		FOSPHOR_AMD_K1H_WAVES=$w python3 bench.py --config C5 --steps ${AB_STEPS:-200} --warmup 20 --no-cpu-baseline --no-other-configs ${AB_ARGS} > gpurun_out/ab/C5_w${w}_$rep.json 2> gpurun_out/ab/C5_w${w}_$rep.err
		python3 tools/bline.py "C5_w${w}_$rep" gpurun_out/ab/C5_w${w}_$rep.json
	done
done
