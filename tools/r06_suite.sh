#!/bin/bash
# the whole GPU suite with per-test durations (the driver's command + --durations)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu --durations=25 2>&1 | tail -45 ) > gpurun_out/r06_suite.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
tail -45 gpurun_out/r06_suite.log | cut -c1-150; tail -2 gpurun_out/r06_smoke.log
