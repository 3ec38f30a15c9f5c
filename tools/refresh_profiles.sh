#!/bin/bash
# Re-create the round's profile artefacts on a GPU box (run through gpurun from the repo root):
#   gpurun_out/rNN_bench_default.log        default bench.py run (the driver's command), JSON line included
#   gpurun_out/rNN_bench_kernel_stats.txt   rocprofv3 --kernel-trace --stats of the same command (fewer steps), per kernel
#   gpurun_out/rNN_prefill_kernel_stats.txt the same for the 1024-token context phase
# then copy them into profiles/ (tracked).  PMC passes: tools/pmc_summary.py.
set -u
R=${1:-r01}
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/${R}_bench_default.log 2>&1
rm -rf gpurun_out/prof_bench gpurun_out/prof_prefill
( cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_bench -- python $ROOT/bench.py --steps 64 --no-cpu-baseline --no-fp16-ref --no-parity --no-prefill --no-batch-sweep --no-tp-prediction ) > gpurun_out/prof_bench.log 2>&1
DB=$(find gpurun_out/prof_bench -name "*_results.db" | head -1)
python tools/rocpd_summary.py "$DB" > gpurun_out/${R}_bench_kernel_stats.txt
( cd /tmp && rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_prefill -- python $ROOT/tools/prefill_probe.py ) > gpurun_out/prof_prefill.log 2>&1
DB=$(find gpurun_out/prof_prefill -name "*_results.db" | head -1)
python tools/rocpd_summary.py "$DB" > gpurun_out/${R}_prefill_kernel_stats.txt
rm -rf gpurun_out/prof_bench gpurun_out/prof_prefill   # the databases are large; the summaries are what is kept
tail -3 gpurun_out/${R}_bench_default.log | cut -c1-600
head -14 gpurun_out/${R}_bench_kernel_stats.txt
