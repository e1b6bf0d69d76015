#!/bin/bash
# Run ON the GPU box: SQ counters of the 8192-point FFT kernel under bench.py --config C3, for one or more prebuilt libraries
# (build/ab/lib_<name>.so; "cur" = the product library).  One rocprofv3 --pmc pass per counter group (kernel trace only).
#   gpurun -- 'bash tools/pmc_c3.sh head cur'
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_c3; mkdir -p $out
for v in "$@"; do
  lib=$PWD/build/ab/lib_$v.so
  [ "$v" = cur ] && lib=$PWD/gr-fosphor_amd/libfosphor_amd.so
  export FOSPHOR_AMD_LIB=$lib
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --kernel-trace -f csv -d $out/k$v/p$i -o p -- python3 bench.py --config C3 --steps 3 --warmup 1 --precondition 0.02 --no-cpu-baseline --no-other-configs > /dev/null 2> $out/k${v}_p$i.log
  done
  python3 tools/pmc_summary.py $(find $out/k$v -name "*counter_collection.csv") | grep -i "k1w\|kernel |" | cut -c1-700 > $out/k$v.md
  cat $out/k$v.md
done
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
