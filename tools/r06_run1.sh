#!/bin/bash
# r06 first contact: the gate|up workgroups of the fused launch - parity, stage clock, A/B
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_fused_qkv_attn.py -x -q -k "gate_up or expired or o_projection_stage_equals" 2>&1 | tail -15 ) > gpurun_out/r06_t1.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_fused_envelope.py -x -q -s 2>&1 | tail -40 ) > gpurun_out/r06_t2.log 2>&1
for d in 0 300 600 900 1200; do
  ( timeout 300 python tools/fused_timeline.py 1024 -1 $d 2>&1 | tail -28 ) > gpurun_out/r06_tl_$d.log 2>&1
done
( timeout 300 python tools/fused_timeline.py 1024 0 2>&1 | tail -22 ) > gpurun_out/r06_tl_off.log 2>&1
B="--no-cpu-baseline --no-parity --no-prefill --no-fp16-ref --no-batch-sweep --steps 128 --warmup 8"
for i in 1 2; do
  for f in "--gemv-gate-up" "" "--mlp-delay 300" "--mlp-delay 900"; do
    timeout 300 python bench.py $B $f > gpurun_out/ab.log 2> gpurun_out/ab.err
    echo "run $i [${f:-default}]: $(tail -1 gpurun_out/ab.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tok/s", round(d["ms_per_step"],4), "ms", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()}, d["step"]["decode_form"], "dominant:", d["roofline"]["kernel"][:60], round(d["roofline"]["frac"],3))' 2>&1 | tail -1)" >> gpurun_out/r06_ab1.txt
  done
done
tail -3 gpurun_out/ab.err >> gpurun_out/r06_ab1.txt
cat gpurun_out/r06_t1.log gpurun_out/r06_t2.log gpurun_out/r06_ab1.txt
