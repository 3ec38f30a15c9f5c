#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_fused_envelope.py -q -m gpu -x -s -k "general or sq_static_pc-kv8" 2>&1 | grep -v amdgpu.ids | tail -30 ) > gpurun_out/r06_t6.log 2>&1
tail -30 gpurun_out/r06_t6.log | cut -c1-200
