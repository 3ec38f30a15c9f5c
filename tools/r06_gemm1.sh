#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_fused_envelope.py -x -q 2>&1 | tail -15 ) > gpurun_out/r06_t2.log 2>&1
for M in 1024 2048 4096; do
  ( timeout 600 python tools/gemm_sweep.py $M 8 42 62 63 70 71 73 2>&1 | tail -12 ) > gpurun_out/r06_gemm_$M.log 2>&1
done
tail -5 gpurun_out/r06_t2.log; cat gpurun_out/r06_gemm_*.log
