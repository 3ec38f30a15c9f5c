"""Decode throughput over the batch size (not a BASELINE configuration: the headline is batch 1): tokens/s and per-layer kernel times
at batch 1 / 2 / 4 / 8 for one configuration, 7B, context 1024, graph replay.  fake_context fills sequence 0 only semantics-free KV
for every sequence (timing only).
    python tools/batch_sweep.py [sq|woq8|woq4|fp16] [context]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import torch  # noqa: E402

import bench  # noqa: E402
from tensorrt_llm.runtime.native import NativeSession  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'sq'
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = dict(bench.LLAMA_7B, num_layers=32)
dev = torch.device('cuda', 0)
int8_kv = mode != 'fp16'
sess = NativeSession(dict(cfg, quant_mode=bench.QM[mode] | (bench.INT8_KV if int8_kv else 0), tp_size=1, tp_rank=0))
w = bench.synth_weights(torch, cfg, mode, int8_kv, 1, 0, dev)
for k, v in w.items():
    sess.set_tensor(k, v)
sess.finalize()
stream = torch.cuda.current_stream().cuda_stream
K = 64
for B in [int(b) for b in os.environ.get("BATCHES", "1,2,4,8").split(",")]:
    sess.setup(B, ctx, 2 * K + 16)
    sess.fake_context(ctx, seed=1, stream=stream)
    sess.step(2, use_graph=False, stream=stream)
    sess.step(4, use_graph=True, stream=stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sess.step(K, use_graph=True, stream=stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = sess.profile(8, stream=stream)
    per = {k: round(v[0] * 1e3 / max(v[1], 1), 2) for k, v in prof.items() if isinstance(v, (list, tuple)) and v[1]}
    print(f'{mode} batch {B}: {dt * 1e3 / K:.3f} ms/step, {B * K / dt:.0f} tokens/s  per-launch us {per}', flush=True)
