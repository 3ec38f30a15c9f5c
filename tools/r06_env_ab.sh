#!/bin/bash
# A/B of HIP runtime environment knobs on one box: bash tools/r06_env_ab.sh "VAR=1" ["VAR2=1" ...]  (each against the plain environment)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
B="--no-cpu-baseline --no-parity --no-prefill --no-fp16-ref --no-batch-sweep --no-tp-prediction --steps 128 --warmup 8"
rm -f gpurun_out/r06_env_ab.txt
for i in 1 2; do
  for e in "" "$@"; do
    env $e timeout 300 python bench.py $B > gpurun_out/ab.log 2> gpurun_out/ab.err
    echo "run $i [${e:-plain}]: $(tail -1 gpurun_out/ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()}, "layer in graph", round(d["step"]["layer_us_in_graph_replay"],2))' 2>&1 | tail -1)" >> gpurun_out/r06_env_ab.txt
  done
done
cat gpurun_out/r06_env_ab.txt
