#!/bin/bash
# Decode tokens/s of every configuration over the context length (the metric is quoted at 1024): one line per (config, context).
#   tools/context_sweep.sh [out=gpurun_out/context_sweep.txt]
out=${1:-gpurun_out/context_sweep.txt}
mkdir -p "$(dirname "$out")"
echo "# bench.py --config C --context L --steps 20 --warmup 5 (decode only; tokens/s, ms per step, HBM fraction of 8 TB/s, attention us per layer)" > "$out"
for c in sq woq8 woq4 fp16; do
  for L in 128 512 1024 2000; do
    python bench.py --config $c --context $L --steps 20 --warmup 5 --no-cpu-baseline --no-prefill --no-fp16-ref --no-parity 2>/dev/null \
      | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c context $L: %.1f tokens/s, %.3f ms/step, hbm %.3f, attention %.2f us/layer, o_proj %.2f' % (d['value'], d['ms_per_step'], d['step']['hbm_frac_of_peak'], d['step']['layer_kernel_us']['attention'], d['step']['layer_kernel_us']['o_proj']))" >> "$out"
  done
done
cat "$out"
