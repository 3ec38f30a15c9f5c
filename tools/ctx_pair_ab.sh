#!/bin/bash
# A/B of the context attention kernel under rocprofv3 (7B geometry): for every S in the argument list the launcher's default,
# the same with two operand stages instead of three (TLLM_CTX_ATTN_NORING=1), paired 64-query blocks forced
# (TLLM_CTX_ATTN_PAIRED=1) and one block per workgroup forced (TLLM_CTX_ATTN_UNPAIRED=1) with three (TLLM_CTX_ATTN_RING=1) and two stages.  Run through gpurun from the repo root.
export TMPDIR=/tmp
R=$(pwd)
for S in ${@:-1024}; do
for v in default noring paired unpaired unpaired_noring default noring paired unpaired unpaired_noring; do
  rm -rf /tmp/pca
  unset TLLM_CTX_ATTN_UNPAIRED TLLM_CTX_ATTN_PAIRED TLLM_CTX_ATTN_NORING TLLM_CTX_ATTN_RING
  case $v in
    noring) export TLLM_CTX_ATTN_NORING=1;;
    paired) export TLLM_CTX_ATTN_PAIRED=1;;
    unpaired) export TLLM_CTX_ATTN_UNPAIRED=1 TLLM_CTX_ATTN_RING=1;;
    unpaired_noring) export TLLM_CTX_ATTN_UNPAIRED=1 TLLM_CTX_ATTN_NORING=1;;
  esac
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pca -- python $R/tools/prefill_probe.py sq $S ) > /tmp/pca.log 2>&1 < /dev/null
  DB=$(find /tmp/pca -name "*_results.db" | head -1)
  echo "S=$S variant $v  ($(grep 'prefill ms' /tmp/pca.log | tail -1))"
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" | grep -E "context_attn" | awk '{print "   ", $1, $2, $3, $4, $8, $9, $10, $11}' | cut -c1-150
done
done
