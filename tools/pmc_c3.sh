#!/bin/bash
export TMPDIR=/tmp
out=$PWD/gpurun_out/r05_pmc_c3; mkdir -p $out
for v in 1 2; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    FOSPHOR_AMD_K1W=$v timeout 200 rocprofv3 --pmc $grp --kernel-trace -f csv -d $out/k$v/p$i -o p -- python3 bench.py --config C3 --steps 3 --warmup 1 --precondition 0.02 --no-cpu-baseline --no-extra-passes > /dev/null 2> $out/k${v}_p$i.log
  done
  python3 tools/pmc_summary.py $(find $out/k$v -name "*counter_collection.csv") | grep -i "k1w\|kernel |" | cut -c1-700 > $out/k$v.md
  cat $out/k$v.md
done
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
