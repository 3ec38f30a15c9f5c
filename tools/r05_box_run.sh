#!/bin/bash
# the headline twice on one fresh box (driver's command line, decode only)  ->  gpurun_out/r05_box_<tag>.txt
set -u
mkdir -p gpurun_out
out=gpurun_out/r05_box_${1:-a}.txt
: > $out
for i in 1 2; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-parity --no-cpu-baseline --no-prefill --no-fp16-ref --no-batch-sweep > gpurun_out/box.log 2>&1 < /dev/null
  tail -1 gpurun_out/box.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tokens/s", round(d["ms_per_step"],4), "ms/step, layer in graph replay", round(d["step"]["layer_us_in_graph_replay"],2), "us")' >> $out
done
cat $out
