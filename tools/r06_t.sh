#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp

( timeout 1200 python tools/prefill_lens.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06_prefill_lens.txt
tail -5 gpurun_out/r06_t4.log | cut -c1-170; cat gpurun_out/r06_prefill_lens.txt
