set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused_qkv_attn.py -x -q -s 2>&1 | tail -60 > gpurun_out/r05_fused_test.log
timeout 900 python -m pytest "tests/test_gpu_bench_geometry.py" -x -q -s -k "sq_static_pc-1-7b or distance" 2>&1 | tail -40 >> gpurun_out/r05_fused_test.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prefill --no-fp16-ref --no-batch-sweep --no-parity > gpurun_out/r05_ab_fused_$i.json 2> gpurun_out/r05_ab_fused_$i.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-prefill --no-fp16-ref --no-batch-sweep --no-parity --two-launch-attention > gpurun_out/r05_ab_two_$i.json 2> gpurun_out/r05_ab_two_$i.err
done
tail -c 600 gpurun_out/r05_ab_*.json
