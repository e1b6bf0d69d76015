#!/bin/bash
# Run ON the GPU box: SQ counters for the kernels of bench.py, one rocprofv3 --pmc pass per group
# (kernels are serialised by counter collection, so K1 is measured without K2/K3 beside it).
#   gpurun -- 'bash tools/pmc_sq.sh r01f'
tag=${1:-rXX}; out=$PWD/gpurun_out/$tag/sq; mkdir -p "$out"; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_VMEM" \
           "GRBM_GUI_ACTIVE SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM"; do
	i=$((i+1))
	rocprofv3 --pmc $grp --kernel-trace -f csv -d "$out/p$i" -o p -- python3 bench.py --steps 2048 --warmup 512 --no-cpu-baseline > /dev/null 2> "$out/p$i.log"
done
python3 tools/pmc_summary.py $(find "$out" -name "*counter_collection.csv") > "$PWD/gpurun_out/$tag/pmc_sq.md"
find "$out" -name "*.csv" -delete; find "$out" -name "*.db" -delete
cat "$PWD/gpurun_out/$tag/pmc_sq.md"
