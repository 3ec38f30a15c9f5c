#!/bin/bash
# A/B of the prefill's RMSNorm (+ quantiser) launch under rocprofv3: register-resident kernel (default) against the LDS-row
# kernel (TLLM_RMSNORM_LDS=1), 7B geometry, S = ${SEQ:-1024}.  Run through gpurun from the repo root.
export TMPDIR=/tmp
R=$(pwd)
for v in reg lds reg lds; do
  rm -rf /tmp/prn
  if [ $v = lds ]; then export TLLM_RMSNORM_LDS=1; else unset TLLM_RMSNORM_LDS; fi
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prn -- python $R/tools/prefill_probe.py sq ${SEQ:-1024} ) > /tmp/prn.log 2>&1 < /dev/null
  DB=$(find /tmp/prn -name "*_results.db" | head -1)
  echo "variant $v: $(grep -i 'prefill' /tmp/prn.log | tail -1 | cut -c1-160)"
  [ -n "$DB" ] && python tools/rocpd_summary.py "$DB" | grep -E "rmsnorm|total kernel" | awk '{print "   ", $1, $2, $3, $4, $8, $9}' | cut -c1-140
done
