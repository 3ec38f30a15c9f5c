#!/bin/bash
# Matrix-pipe evidence for the SmoothQuant prefill GEMM (north_star: "MFMA utilisation against gfx950 peak"):
#   gpurun_out/rNN_mfma_ceiling.txt   what the int8 matrix pipe of this chip sustains (csrc/tools/mfma_probe.cpp)
#   gpurun_out/rNN_mfma_gemm.txt      the four 7B shapes at M = 1024, un-profiled (tools/gemm_probe.py)
#   gpurun_out/rNN_mfma_pmc.txt       SQ / GRBM counters per GEMM kernel (separate passes: 8 SQ slots, 2 GRBM)
# Only --kernel-trace next to --pmc (gpurun refuses other trace domains with counters).  Copy the three into profiles/.
set -u
R=${1:-r02}
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
trtllm-llama_amd/csrc/build/mfma_probe > gpurun_out/${R}_mfma_ceiling.txt 2>&1
python tools/gemm_probe.py > gpurun_out/${R}_mfma_gemm.txt 2>&1
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_mfma_$i
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $set -d $ROOT/gpurun_out/pmc_mfma_$i -o pmc -- python $ROOT/tools/gemm_probe.py ) > gpurun_out/pmc_mfma_$i.log 2>&1
done
python tools/mfma_pmc_summary.py $(find gpurun_out/pmc_mfma_1 gpurun_out/pmc_mfma_2 -name "*_results.db") > gpurun_out/${R}_mfma_pmc.txt 2>&1
rm -rf gpurun_out/pmc_mfma_1 gpurun_out/pmc_mfma_2
cat gpurun_out/${R}_mfma_ceiling.txt; cat gpurun_out/${R}_mfma_gemm.txt; head -60 gpurun_out/${R}_mfma_pmc.txt
