#!/bin/bash
# Issue counters of the decode GEMV kernels of one configuration (default woq4): what the waves do while the weights stream.
# Two rocprofv3 --pmc passes (--kernel-trace only next to them), summarised on the box.
#   bash tools/gemv_pmc.sh [config]      -> gpurun_out/gemv_pmc_<config>.txt
#   CMD='<command>' bash tools/gemv_pmc.sh <tag>   -> the same counters under another command (tools/batch_sweep.py did NOT finish under
#   the counter passes within 7 minutes on r04's box - its set-up work is collected too; give it a command that only runs steps)
set -u
CFG=${1:-woq4}
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf gpurun_out/gpmc_$i
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $P -d $ROOT/gpurun_out/gpmc_$i -o pmc -- ${CMD:-python $ROOT/bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline --no-fp16-ref --no-prefill --no-parity} ) > gpurun_out/gpmc_$i.log 2>&1
done
python tools/pmc_kernel_summary.py 'gemv_kernel|gemv_ksplit|gemv_mfma' $(find gpurun_out/gpmc_1 gpurun_out/gpmc_2 -name "*_results.db") > gpurun_out/gemv_pmc_$CFG.txt
rm -rf gpurun_out/gpmc_1 gpurun_out/gpmc_2
cat gpurun_out/gemv_pmc_$CFG.txt | cut -c1-180
