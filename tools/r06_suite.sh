#!/bin/bash
# the GPU suite with per-test durations + a default bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -q -m gpu --durations=60 -x 2>&1 | tail -90 ) > gpurun_out/r06_suite.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-parity --no-batch-sweep > gpurun_out/r06_bench_quick.json 2> gpurun_out/r06_bench_quick.err
tail -80 gpurun_out/r06_suite.log; python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_quick.json').readline()); print(d['value'], d['roofline']['kernel'], d['roofline']['frac']); print([ (k['kernel'][:30], round(k['avg_launch_us'],2), round(k['frac'],3), k['traffic']) for k in d['roofline']['kernels']])"
