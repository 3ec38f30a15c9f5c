#!/bin/bash
# A/B on one box, weight-only int8 decode (BASELINE.json configs[2]): the fused QKV + attention launch against the two launches,
# alternating, three runs each.   tools/r05_woq8_ab.sh  ->  gpurun_out/r05_woq8_fused_ab.txt
set -u
mkdir -p gpurun_out
out=gpurun_out/r05_woq8_fused_ab.txt
: > $out
for i in 1 2 3; do
  for f in "" "--two-launch-attention"; do
    timeout 300 python bench.py --config woq8 --steps 128 --warmup 8 --no-cpu-baseline --no-prefill --no-fp16-ref --no-batch-sweep --no-parity $f > gpurun_out/ab.log 2>&1 < /dev/null
    echo "run $i woq8 ${f:-fused}: $(tail -1 gpurun_out/ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()})')" >> $out
  done
done
cat $out
