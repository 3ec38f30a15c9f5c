#!/bin/bash
# r06 evidence on one GPU box: driver-cmdline bench, default bench, rocprof kernel stats (decode + prefill), PMC passes (sq, fp16)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_bench_driver_cmdline.log 2>&1
tail -1 gpurun_out/r06_bench_driver_cmdline.log | head -c 300; echo
timeout 900 bash tools/refresh_profiles.sh r06 > gpurun_out/refresh.log 2>&1 < /dev/null
timeout 600 bash tools/refresh_pmc.sh r06 sq > gpurun_out/refresh_pmc.log 2>&1 < /dev/null
timeout 600 bash tools/refresh_pmc.sh r06 fp16 > gpurun_out/refresh_pmc16.log 2>&1 < /dev/null
head -8 gpurun_out/r06_bench_kernel_stats.txt | cut -c1-200
head -6 gpurun_out/r06_pmc_sq.txt | cut -c1-200
