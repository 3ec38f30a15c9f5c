#!/bin/bash
# the headline twice on this box (driver command line, side reports off) -> one line appended to gpurun_out/r06_box_spread.txt
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-batch-sweep --no-tp-prediction --no-prefill --no-fp16-ref"
r=""
for i in 1 2; do
  timeout 300 python bench.py $B > gpurun_out/bx.log 2> gpurun_out/bx.err
  r="$r $(tail -1 gpurun_out/bx.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s (layer in replay %.2f us, front %.2f gate|up %.2f down %.2f)" % (d["step"]["layer_us_in_graph_replay"], d["step"]["layer_kernel_us"]["front"], d["step"]["layer_kernel_us"]["gate_up"], d["step"]["layer_kernel_us"]["down"]))' 2>&1 | tail -1) |"
done
echo "box $(hostname) $(date -u +%H:%M):$r" | tee gpurun_out/r06_box_line.txt
