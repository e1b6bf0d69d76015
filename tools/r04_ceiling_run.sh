#!/bin/bash
# Run ON the GPU box (gpurun -- 'bash tools/r04_ceiling_run.sh'): the ceiling probes of DESIGN_HISTORY.md 4a, raw bench lines kept.
# Every line is `bench.py --steps 40 --warmup 8 --no-cpu-baseline` on C2 (256 bins, batch mode), two repetitions each.
OUT=gpurun_out/r04_ceiling
mkdir -p $OUT
run() {		# label, lib ('' = product), env...
	local label=$1 lib=$2; shift 2
	for rep in 1 2; do
		if [ -n "$lib" ]; then
			env FOSPHOR_AMD_LIB=$PWD/build/ab/lib_$lib.so "$@" python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > $OUT/${label}_$rep.json 2> $OUT/${label}_$rep.err
		else
			env "$@" python3 bench.py --steps 40 --warmup 8 --no-cpu-baseline > $OUT/${label}_$rep.json 2> $OUT/${label}_$rep.err
		fi
		python3 tools/bline.py "$label" $OUT/${label}_$rep.json
	done
}
run full            ""       X=1
run k1_only         probes     FOSPHOR_AMD_DBG_SKIP=2
run k1_k3_only      probes     FOSPHOR_AMD_DBG_SKIP=8
run k1_k2_only      probes     FOSPHOR_AMD_DBG_SKIP=4
run k1_only_nobins  nobins   FOSPHOR_AMD_DBG_SKIP=2
run k1_only_ldsatom ldsatom  FOSPHOR_AMD_DBG_SKIP=2
run k1_only_epi15   epi15    FOSPHOR_AMD_DBG_SKIP=2
run full_k2noatom  k2noatom X=1
run full_k2store   k2store  X=1
run cumask16        probes     FOSPHOR_AMD_DBG_CUMASK=16
run cumask32        probes     FOSPHOR_AMD_DBG_CUMASK=32
run cumask48        probes     FOSPHOR_AMD_DBG_CUMASK=48
echo "--- read_skew (K1's loads alone / loads + stores, two launches in flight)"
./tools/ubench/read_skew 2>&1 | tee $OUT/read_skew.txt
echo "--- hbm_ceiling.py"
python3 tools/ubench/hbm_ceiling.py 2>&1 | tee $OUT/hbm_ceiling.txt
