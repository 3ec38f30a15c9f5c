#!/bin/bash
L=trtllm-llama_amd/tensorrt_llm/libs/libnvinfer_plugin_tensorrt_llm.so
for i in $(seq 1 ${2:-2}); do
  for v in ${3:-baseline new}; do
    cp abtest/$v.so $L
    timeout 200 python bench.py --config ${1:-sq} --no-cpu-baseline --no-prefill --no-fp16-ref --steps 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $v', round(d['value'],1), round(d['ms_per_step'],4), {k: round(x,2) for k,x in d['step']['layer_kernel_us'].items()}, round(d['step']['profile_ms_per_step']['gemv_head']*1000,1))"
  done
done
cp abtest/new.so $L
