#!/bin/bash
# M0 save/restore A/B (GEMM sweep + decode bench), then the whole GPU suite with per-test durations
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tools/gemm_sweep.py 1024 8 42 62 63 20 2>&1 | tail -6 ) > gpurun_out/r06_gemm_m0.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-parity --no-batch-sweep > gpurun_out/r06_bench_quick.json 2> gpurun_out/r06_bench_quick.err
( time timeout 1500 python -m pytest tests -q -m gpu --durations=80 2>&1 | tail -110 ) > gpurun_out/r06_suite.log 2>&1
cat gpurun_out/r06_gemm_m0.log | cut -c1-215; python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_quick.json').readline()); print(d['value'], d.get('prefill',{}).get('ms'), d.get('fp16_tokens_per_s'), d.get('woq8_tokens_per_s')); print([ (k['kernel'][:30], round(k['avg_launch_us'],2), round(k['frac'],3)) for k in d['roofline']['kernels']]); print({k:(round(v['us'],1), round(v['frac_of_5POPs'],3), v['tactic']) for k,v in d['sq_gemm_mfma'].items()})"
tail -100 gpurun_out/r06_suite.log | cut -c1-150
