#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 800 bash tools/refresh_pmc.sh r06 sq > gpurun_out/refresh_pmc.log 2>&1 < /dev/null
timeout 800 bash tools/refresh_pmc.sh r06 fp16 > gpurun_out/refresh_pmc16.log 2>&1 < /dev/null
head -12 gpurun_out/r06_pmc_sq.txt | cut -c1-200
head -8 gpurun_out/r06_pmc_fp16.txt | cut -c1-200
