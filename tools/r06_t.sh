#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_mlp_fused.py -q -m gpu -x 2>&1 | tail -30 ) > gpurun_out/r06_t5.log 2>&1
( timeout 300 python tools/fused_timeline.py 1024 fuse_mlp=1 2>&1 | grep -v amdgpu.ids | tail -12 ) > gpurun_out/r06_mlp_tl.txt 2>&1
bash tools/r06_ab.sh "" "--one-launch-mlp" 2 > /dev/null 2>&1
tail -4 gpurun_out/r06_t5.log | cut -c1-200; cat gpurun_out/r06_mlp_tl.txt; cat gpurun_out/r06_ab.txt
