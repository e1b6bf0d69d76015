#!/bin/bash
# Run ON the GPU box: the cache policy of the 65536-point kernel's IQ requests (build/ab/lib_iq<k>.so from tools/ab_build.sh "iq1:-DK1H_IQ_POL=1" ...;
# cur = nt): path rate and the FFT kernel's WRITE_SIZE / FETCH_SIZE per frame.
export TMPDIR=/tmp
mkdir -p gpurun_out/ab
for n in cur "$@"; do
	lib=$PWD/build/ab/lib_$n.so; [ "$n" = cur ] && lib=$PWD/gr-fosphor_amd/libfosphor_amd.so
	FOSPHOR_AMD_LIB=$lib python3 bench.py --config C5 --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs > gpurun_out/ab/s.json 2>/dev/null
	python3 tools/bline.py "C5 $n" gpurun_out/ab/s.json
	for c in WRITE_SIZE FETCH_SIZE; do
		rm -rf /tmp/pm; FOSPHOR_AMD_LIB=$lib timeout 200 rocprofv3 --pmc $c --kernel-trace -f csv -d /tmp/pm -o p -- python3 bench.py --config C5 --steps 4 --warmup 2 --precondition 0.05 --no-cpu-baseline --no-traffic-twin --no-extra-passes --no-other-configs > /dev/null 2>&1
		python3 tools/pmc_summary.py $(find /tmp/pm -name "*counter_collection.csv") 2>/dev/null | grep k1h_fused | cut -c1-140
	done
done
