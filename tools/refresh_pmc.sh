#!/bin/bash
# HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they cannot share a pass on gfx950),
# summarised on the GPU box (the counter databases are too large to travel):
#   gpurun_out/rNN_pmc_<config>.txt, gpurun_out/rNN_pmc_summary.json  -> copy into profiles/
# Only --kernel-trace next to --pmc (gpurun refuses other trace domains with counters).
set -u
R=${1:-r01}; CFG=${2:-sq}
ROOT=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $c -d $ROOT/gpurun_out/pmc_$c -o pmc -- python $ROOT/bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline --no-fp16-ref --no-prefill --no-parity --no-batch-sweep --no-tp-prediction ) > gpurun_out/pmc_$c.log 2>&1
done
F=$(find gpurun_out/pmc_FETCH_SIZE -name "*_results.db" | head -1); W=$(find gpurun_out/pmc_WRITE_SIZE -name "*_results.db" | head -1)
cp profiles/${R}_pmc_summary.json gpurun_out/${R}_pmc_summary.json 2>/dev/null
python tools/pmc_summary.py "$F" "$W" $CFG gpurun_out/${R}_pmc_summary.json > gpurun_out/${R}_pmc_$CFG.txt
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
cat gpurun_out/${R}_pmc_$CFG.txt | cut -c1-170
