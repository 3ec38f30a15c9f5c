#!/bin/bash
# A/B on one box: the O-projection as the third stage of the fused launch against the GEMV launch (bench.py --gemv-o-projection),
# alternating, three runs each.   tools/r05_o_ab.sh  ->  gpurun_out/r05_o_stage_ab.txt
set -u
mkdir -p gpurun_out
out=gpurun_out/r05_o_stage_ab.txt
: > $out
for i in 1 2 3; do
  for f in "" "--gemv-o-projection"; do
    timeout 300 python bench.py --steps 128 --warmup 8 --no-cpu-baseline --no-prefill --no-fp16-ref --no-batch-sweep --no-parity $f > gpurun_out/ab.log 2>&1
    echo "run $i ${f:-fused-o}: $(tail -1 gpurun_out/ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()})')" >> $out
  done
done
cat $out
