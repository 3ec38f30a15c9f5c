#!/bin/bash
# A/B on one box, weight-only int8 decode: the O-projection stage of the fused launch against the GEMV launch, alternating.
#   tools/r05_woq8_o_ab.sh  ->  gpurun_out/r05_woq8_o_stage_ab.txt
set -u
mkdir -p gpurun_out
out=gpurun_out/r05_woq8_o_stage_ab.txt
: > $out
for i in 1 2 3; do
  for f in "" "--gemv-o-projection"; do
    timeout 300 python bench.py --config woq8 --steps 128 --warmup 8 --no-cpu-baseline --no-prefill --no-fp16-ref --no-batch-sweep --no-parity $f > gpurun_out/ab.log 2>&1 < /dev/null
    echo "run $i woq8 ${f:-three-stage launch}: $(tail -1 gpurun_out/ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms, layer in graph replay", round(d["step"]["layer_us_in_graph_replay"],2), "us")')" >> $out
  done
done
cat $out
