#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
( cd trtllm-llama_amd/csrc && for i in 1 2; do timeout 300 build/microbench 0 0 2>&1 | grep -E "^sq   (qkv|gateup)|x sq N12288 K4096 none|x sq gateup" ; done ) > gpurun_out/r06_prologue_cost.txt 2>&1
cat gpurun_out/r06_prologue_cost.txt
