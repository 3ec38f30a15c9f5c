#!/bin/bash
# per-kernel stats of the 1024-token SmoothQuant prefill (tactic profile off: the static rule's kernels, no profiling launches in the trace)
#   tools/prefill_stats.sh [config] -> gpurun_out/prefill_stats.txt
set -u
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/prof_prefill
( cd /tmp && TLLM_GEMM_TACTICS=off timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_prefill -- python $ROOT/tools/prefill_probe.py ${1:-sq} ) > gpurun_out/prof_prefill.log 2>&1 < /dev/null
DB=$(find gpurun_out/prof_prefill -name "*_results.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_summary.py "$DB" | head -24 | cut -c1-230 > gpurun_out/prefill_stats.txt; else echo "no trace" > gpurun_out/prefill_stats.txt; fi
rm -rf gpurun_out/prof_prefill
cat gpurun_out/prefill_stats.txt; tail -3 gpurun_out/prof_prefill.log | cut -c1-300
