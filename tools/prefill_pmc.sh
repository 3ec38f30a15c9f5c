#!/bin/bash
# SQ / GRBM counters of the prefill's non-GEMM kernels (context attention, RoPE + KV write + V^T, RMSNorm) at the 7B geometry,
# S = ${SEQ:-1024}: gpurun_out/rNN_prefill_pmc.txt -> copy into profiles/.  Separate passes (8 SQ slots); only --kernel-trace
# next to --pmc (gpurun refuses other trace domains with counters).  Run through gpurun from the repo root.
set -u
R=${1:-r02}
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/pmc_pf_$i
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $ROOT/gpurun_out/pmc_pf_$i -o pmc -- python $ROOT/tools/prefill_probe.py sq ${SEQ:-1024} ) > gpurun_out/pmc_pf_$i.log 2>&1
done
python tools/pmc_kernel_summary.py 'context_attn|rope_kv|rmsnorm' $(find gpurun_out/pmc_pf_1 gpurun_out/pmc_pf_2 gpurun_out/pmc_pf_3 -name "*_results.db") > gpurun_out/${R}_prefill_pmc.txt 2>&1
rm -rf gpurun_out/pmc_pf_1 gpurun_out/pmc_pf_2 gpurun_out/pmc_pf_3
cat gpurun_out/${R}_prefill_pmc.txt
