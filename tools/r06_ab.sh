#!/bin/bash
# A/B of a session-key variant on one box: bash tools/r06_ab.sh "<flags A>" "<flags B>" [runs]
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
B="--no-cpu-baseline --no-parity --no-prefill --no-fp16-ref --no-batch-sweep --no-tp-prediction --steps 128 --warmup 8"
rm -f gpurun_out/r06_ab.txt
for i in $(seq 1 ${3:-3}); do
  for f in "$1" "$2"; do
    timeout 300 python bench.py $B $f > gpurun_out/ab.log 2> gpurun_out/ab.err
    echo "run $i [${f:-default}]: $(tail -1 gpurun_out/ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()})' 2>&1 | tail -1)" >> gpurun_out/r06_ab.txt
  done
done
cat gpurun_out/r06_ab.txt
