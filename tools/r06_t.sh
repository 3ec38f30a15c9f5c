#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests/test_gpu_fused_qkv_attn.py tests/test_gpu_fused_envelope.py tests/test_gpu_plugins.py -q -m gpu -x -k "woq4 or weight_only or o_projection" 2>&1 | tail -12 ) > gpurun_out/r06_t3.log 2>&1
B="--no-cpu-baseline --no-parity --no-prefill --no-fp16-ref --no-batch-sweep --no-tp-prediction --steps 128 --warmup 8"
rm -f gpurun_out/r06_ab3.txt
for i in 1 2; do
  for f in "--config woq4 --two-launch-attention" "--config woq4"; do
    timeout 300 python bench.py $B $f > gpurun_out/ab.log 2> gpurun_out/ab.err
    echo "run $i [$f]: $(tail -1 gpurun_out/ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()}, d["roofline"]["kernel"][:40], round(d["roofline"]["frac"],3))' 2>&1 | tail -1)" >> gpurun_out/r06_ab3.txt
  done
done
tail -12 gpurun_out/r06_t3.log | cut -c1-170; cat gpurun_out/r06_ab3.txt
