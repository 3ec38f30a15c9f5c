set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prefill --no-fp16-ref --no-batch-sweep --no-parity > gpurun_out/r05_ab_fused_$i.json 2> gpurun_out/r05_ab_fused_$i.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prefill --no-fp16-ref --no-batch-sweep --no-parity --two-launch-attention > gpurun_out/r05_ab_two_$i.json 2> gpurun_out/r05_ab_two_$i.err
done
grep -o '"value": [0-9.]*\|"layer_kernel_us": {[^}]*}' gpurun_out/r05_ab_*.json
timeout 1500 python -m pytest tests/test_gpu_full_size.py tests/test_gpu_bench_geometry.py -x -q 2>&1 | tail -30 > gpurun_out/r05_fullsize.log
tail -30 gpurun_out/r05_fullsize.log
