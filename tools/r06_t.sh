#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/fused_timeline.py 1024 2>&1 | grep -v amdgpu.ids | tail -22; echo "=== early"; timeout 300 python tools/fused_timeline.py 1024 fused_early_attention=1 2>&1 | grep -v amdgpu.ids | tail -22 ) > gpurun_out/r06_early_tl.txt 2>&1
bash tools/r06_ab.sh "" "--session-key fused_early_attention=1" 2 > /dev/null 2>&1
cat gpurun_out/r06_early_tl.txt gpurun_out/r06_ab.txt
