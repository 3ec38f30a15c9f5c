#!/bin/bash
# the other BASELINE configurations at the driver's command line (decode + 1024-token prefill), and the decode over the context length
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/r06_other_configs.txt; rm -f $out
B="--gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-fp16-ref --no-batch-sweep --no-tp-prediction"
for c in sq fp16 woq8 woq4; do
  timeout 400 python bench.py $B --config $c > gpurun_out/oc.log 2> gpurun_out/oc.err
  echo "$c: $(tail -1 gpurun_out/oc.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tokens/s,", round(d["ms_per_step"],4), "ms/step, prefill", round(d["prefill"]["ms"],2), "ms, layer launches (us)", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()}, "dominant", d["roofline"]["kernel"][:36], round(d["roofline"]["frac"],3), "step hbm", round(d["step"]["hbm_frac_of_peak"],3))' 2>&1 | tail -1)" >> $out
done
for L in 128 512 1024 2000; do
  timeout 300 python bench.py $B --no-prefill --context $L > gpurun_out/oc.log 2> gpurun_out/oc.err
  echo "sq context $L: $(tail -1 gpurun_out/oc.log | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["value"],1), "tokens/s,", round(d["ms_per_step"],4), "ms/step", {k: round(v,2) for k,v in d["step"]["layer_kernel_us"].items()})' 2>&1 | tail -1)" >> $out
done
cat $out
