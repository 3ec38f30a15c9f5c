#!/bin/bash
# r05 evidence on one GPU box: default bench, driver-cmdline bench, rocprof kernel stats (decode + prefill), PMC passes (sq)
set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r05_bench_driver_cmdline.log 2>&1
tail -1 gpurun_out/r05_bench_driver_cmdline.log | head -c 300; echo
bash tools/refresh_profiles.sh r05 > gpurun_out/refresh.log 2>&1 < /dev/null
bash tools/refresh_pmc.sh r05 sq > gpurun_out/refresh_pmc.log 2>&1 < /dev/null
head -20 gpurun_out/r05_bench_kernel_stats.txt
head -12 gpurun_out/r05_pmc_sq.txt
