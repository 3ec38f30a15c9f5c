#!/bin/bash
L=trtllm-llama_amd/tensorrt_llm/libs/libnvinfer_plugin_tensorrt_llm.so
for i in 1 2; do for v in baseline new; do cp abtest/$v.so $L; timeout 200 python bench.py --no-cpu-baseline --no-fp16-ref --steps 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v prefill ms', round(d['prefill']['ms'],3))"; done; done
cp abtest/new.so $L
