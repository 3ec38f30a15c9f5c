#!/bin/bash
# Kernel times of the context attention path (S = ${SEQ:-1024}, 7B geometry) under rocprofv3: run through gpurun from the repo root.
# Arguments are labels "NQ:ABL" (TLLM_CTX_ATTN_NQ = 4 | 2 query slices per workgroup; TLLM_CTX_ATTN_ABL a compiled-in timing variant
# while one exists); other A/B switches (TLLM_CTX_ATTN_OLD, TLLM_CTX_ATTN_UNFUSED_VT) come from the environment.
export TMPDIR=/tmp
R=$(pwd)
for cfg in "$@"; do
  nq=${cfg%%:*}; abl=${cfg##*:}
  rm -rf /tmp/pca
  ( cd /tmp && TLLM_CTX_ATTN_NQ=$nq TLLM_CTX_ATTN_ABL=$abl timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pca -- python $R/tools/prefill_probe.py sq ${SEQ:-1024} ) > /dev/null 2>&1 < /dev/null
  DB=$(find /tmp/pca -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then
    echo "NQ=$nq ABL=$abl"; python tools/rocpd_summary.py "$DB" | grep -E "context_attn|rope_kv|v_transpose" | awk '{print "   ", $2, $3, $4, $8}' | cut -c1-120
  else
    echo "NQ=$nq ABL=$abl: no profile"
  fi
done
