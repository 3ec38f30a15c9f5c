import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import torch, numpy as np
import bench
from tensorrt_llm.runtime.native import NativeSession
class A: pass
mode = sys.argv[1] if len(sys.argv) > 1 else 'sq'
cfg = dict(bench.LLAMA_7B, num_layers=32)
dev = torch.device('cuda', 0)
int8_kv = mode != 'fp16'
qm = bench.QM[mode] | (bench.INT8_KV if int8_kv else 0)
sess = NativeSession(dict(cfg, quant_mode=qm, tp_size=1, tp_rank=0))
w = bench.synth_weights(torch, cfg, mode, int8_kv, 1, 0, dev)
for k, v in w.items(): sess.set_tensor(k, v)
sess.finalize()
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
sess.setup(1, S, 8)
ids = np.random.default_rng(1).integers(3, 32000, (1, S)).astype(np.int32)
lens = np.array([S], np.int32)
stream = torch.cuda.current_stream().cuda_stream
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sess.context(ids, lens, stream=stream)
    torch.cuda.synchronize(); print(mode, 'prefill ms', (time.perf_counter() - t0) * 1e3)
