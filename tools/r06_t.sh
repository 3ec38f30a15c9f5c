#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_plugins.py -q -x -k "split_k" 2>&1 | tail -15 ) > gpurun_out/r06_t4.log 2>&1
( SHAPES="4096,4096;4096,11008" timeout 600 python tools/gemm_sweep.py 1024 8 42 62 64 2>&1 | tail -6 ) > gpurun_out/r06_splitk_sq.log 2>&1
( SHAPES="4096,4096;4096,11008" timeout 600 python tools/gemm_sweep.py 512 8 42 62 64 2>&1 | tail -6 ) >> gpurun_out/r06_splitk_sq.log 2>&1
( SHAPES="4096,4096;4096,11008" timeout 600 python tools/fp16_gemm_sweep.py 1024 8 54 56 57 2>&1 | tail -6 ) > gpurun_out/r06_splitk_f16.log 2>&1
tail -15 gpurun_out/r06_t4.log | cut -c1-200; cat gpurun_out/r06_splitk_sq.log gpurun_out/r06_splitk_f16.log | cut -c1-200
