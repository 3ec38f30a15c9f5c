#!/bin/bash
# SQ counters of the SmoothQuant GEMM tile shapes / ablations on one problem (default QKV at M = 1024):
#   [M=4096] tools/sqp_pmc.sh "<cfg ids>" ["N,K"]  ->  gpurun_out/sqp_pmc.txt
set -u
CFGS=${1:-"6 13"}
export SHAPES=${2:-"12288,4096"}
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_sqp_$i
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $ROOT/gpurun_out/pmc_sqp_$i -o pmc -- python $ROOT/tools/gemm_sweep.py ${M:-1024} $CFGS ) > gpurun_out/pmc_sqp_$i.log 2>&1
done
python tools/mfma_pmc_summary.py $(find gpurun_out/pmc_sqp_1 gpurun_out/pmc_sqp_2 -name "*_results.db") > gpurun_out/sqp_pmc.txt 2>&1
rm -rf gpurun_out/pmc_sqp_1 gpurun_out/pmc_sqp_2
cat gpurun_out/sqp_pmc.txt
