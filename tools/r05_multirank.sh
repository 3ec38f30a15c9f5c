#!/bin/bash
# the driver's multi-rank launch lines on ONE GPU (ranks share it; TLLM_TEST_SHARED_GPU=1: gloo for
# torch.distributed, the peer-to-peer transport for the data path).   tools/r05_multirank.sh  ->  gpurun_out/r05_multirank.txt
set -u
export TLLM_TEST_SHARED_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
out=gpurun_out/r05_multirank.txt
: > $out
run() {
  echo "== $*" >> $out
  timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus $1 ${@:3} > gpurun_out/mr_$1.log 2>&1
  echo "rc=$?" >> $out
  tail -1 gpurun_out/mr_$1.log | cut -c1-1800 >> $out
}
run 2 29511 --steps 16 --warmup 4 --no-prefill
run 4 29512 --steps 16 --warmup 4 --no-prefill
run 8 29513 --steps 16 --warmup 4 --layers 4 --no-prefill
cat $out
