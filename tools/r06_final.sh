#!/bin/bash
# r06 final evidence on one GPU box (no PMC passes: the decode kernels are those of tools/r06_profiles.sh's run)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r06_bench_driver_cmdline.log 2>&1
tail -1 gpurun_out/r06_bench_driver_cmdline.log | head -c 300; echo
timeout 1200 bash tools/refresh_profiles.sh r06 > gpurun_out/refresh.log 2>&1 < /dev/null
head -8 gpurun_out/r06_bench_kernel_stats.txt | cut -c1-200
tail -2 gpurun_out/r06_bench_default.log | head -c 400
