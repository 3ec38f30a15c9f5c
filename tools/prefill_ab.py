"""1024-token prefill, alternating between session keys on one box (same weights):  python tools/prefill_ab.py key=value [key=value ...]
Each key=value is a variant next to the default; e.g.  python tools/prefill_ab.py dual_mlp_gemm=0"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import torch, numpy as np
import bench
from tensorrt_llm.runtime.native import NativeSession
cfg = dict(bench.LLAMA_7B, num_layers=32)
dev = torch.device('cuda', 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
qm = bench.QM['sq'] | bench.INT8_KV
w = bench.synth_weights(torch, cfg, 'sq', True, 1, 0, dev)
S = 1024
ids = np.random.default_rng(1).integers(3, 32000, (1, S)).astype(np.int32)
lens = np.array([S], np.int32)
stream = torch.cuda.current_stream().cuda_stream
variants = [dict()] + [dict([a.split('=')]) for a in sys.argv[1:]]
sessions = []
for v in variants:
    s = NativeSession(dict(cfg, quant_mode=qm, tp_size=1, tp_rank=0, **{k: int(x) for k, x in v.items()}))
    for k, t in w.items():
        s.set_tensor(k, t)
    s.finalize()
    s.setup(1, S, 8)
    s.context(ids, lens, stream=stream)
    sessions.append(s)
logits = [s.logits(stream=stream) for s in sessions]
for i, v in enumerate(variants[1:], 1):
    print(v, 'context logits max |d| vs default:', float(np.abs(logits[i] - logits[0]).max()))
for rnd in range(4):
    for v, s in zip(variants, sessions):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            s.context(ids, lens, stream=stream)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(f'round {rnd} {v or "default"}: prefill {min(ts):.3f} ms')
