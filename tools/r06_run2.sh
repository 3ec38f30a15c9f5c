#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_fused_envelope.py -x -q -s 2>&1 | tail -40 ) > gpurun_out/r06_t2.log 2>&1
for cfg in "1200 1" "1500 1" "1500 2" "1700 2" "0 0" ; do
  set -- $cfg
  ( timeout 300 python tools/fused_timeline.py 1024 -1 $1 $2 2>&1 | tail -24 | head -22 ) > gpurun_out/r06_tl_$1_$2.log 2>&1
done
B="--no-cpu-baseline --no-parity --no-prefill --no-fp16-ref --no-batch-sweep --steps 128 --warmup 8"
rm -f gpurun_out/r06_ab2.txt
for i in 1 2; do
  for f in "--gemv-gate-up" "--mlp-delay 1200 --mlp-tiles 1" "--mlp-delay 1500 --mlp-tiles 1" "--mlp-delay 1500 --mlp-tiles 2" "--mlp-delay 1700 --mlp-tiles 2" "--mlp-tiles 0"; do
    timeout 300 python bench.py $B $f > gpurun_out/ab.log 2> gpurun_out/ab.err
    echo "run $i [${f:-default}]: $(tail -1 gpurun_out/ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()})' 2>&1 | tail -1)" >> gpurun_out/r06_ab2.txt
  done
done
tail -30 gpurun_out/r06_t2.log; cat gpurun_out/r06_ab2.txt
