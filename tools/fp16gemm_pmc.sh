#!/bin/bash
# Matrix-pipe / LDS counters of the fp16 and weight-only prefill GEMMs at the four 7B shapes, M = 1024 (north_star: "MFMA (i32 int8
# and fp16) ... evidenced by rocprof ... MFMA utilisation"):   tools/fp16gemm_pmc.sh [rNN]  ->  gpurun_out/rNN_fp16gemm_pmc.txt
set -u
R=${1:-r04}
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_f16_$i
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $ROOT/gpurun_out/pmc_f16_$i -o pmc -- python $ROOT/tools/woq_gemm_sweep.py 1024 ) > gpurun_out/pmc_f16_$i.log 2>&1
done
{ echo "# un-profiled timings of the same process (tools/woq_gemm_sweep.py 1024):"; python tools/woq_gemm_sweep.py 1024;
  echo; python tools/mfma_pmc_summary.py $(find gpurun_out/pmc_f16_1 gpurun_out/pmc_f16_2 -name "*_results.db"); } > gpurun_out/${R}_fp16gemm_pmc.txt 2>&1
rm -rf gpurun_out/pmc_f16_1 gpurun_out/pmc_f16_2
tail -60 gpurun_out/${R}_fp16gemm_pmc.txt
