#!/bin/bash
out=gpurun_out/ab3; mkdir -p $out
export TMPDIR=/tmp
FOSPHOR_AMD_K1=6 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
b() { label=$1; shift; env "$@" python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-traffic-twin 2>$out/$label.err | python3 tools/bline.py $label; }
b k1_k23 FOSPHOR_AMD_K1=1
b k6_k23 FOSPHOR_AMD_K1=6
b k6_k23_again FOSPHOR_AMD_K1=6
b k5_k23 FOSPHOR_AMD_K1=5
b k6_nok23 FOSPHOR_AMD_K1=6 FOSPHOR_AMD_K23=0
b k1_nok23 FOSPHOR_AMD_K1=1 FOSPHOR_AMD_K23=0
b k6_k23_tile32 FOSPHOR_AMD_K1=6 FOSPHOR_AMD_TILE=32
FOSPHOR_AMD_K1=6 rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d $out/pmcF -o p -- python3 bench.py --steps 8 --warmup 4 --precondition 0.05 --no-cpu-baseline --no-traffic-twin --no-extra-passes > /dev/null 2> $out/pmcF.log
FOSPHOR_AMD_K1=6 rocprofv3 --pmc WRITE_SIZE --kernel-trace -f csv -d $out/pmcW -o p -- python3 bench.py --steps 8 --warmup 4 --precondition 0.05 --no-cpu-baseline --no-traffic-twin --no-extra-passes > /dev/null 2> $out/pmcW.log
python3 tools/pmc_summary.py $(find $out/pmcF $out/pmcW -name "*counter_collection.csv") 2>&1 | tail -12
find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete
