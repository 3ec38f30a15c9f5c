"""Prefill at several input lengths, SmoothQuant / fp16 / weight-only int8 engines: the tactic table as profiled (with the split-K
128 x 128 ids 65 / 58) against the same table with those ids mapped back to 64 / 57 (r06 A/B on one box, alternating).
    python tools/prefill_lens.py [mode ...]     modes: sq fp16 woq8"""
import ctypes, sys, time, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import torch, numpy as np
import bench
from tensorrt_llm.plugin import capi
from tensorrt_llm.runtime.native import NativeSession
lib = capi.load_library()
lib.tllm_gemm_tactics_export.argtypes = [ctypes.c_char_p, ctypes.c_int64]
lib.tllm_gemm_tactics_export.restype = ctypes.c_int64
lib.tllm_gemm_tactics_import.argtypes = [ctypes.c_char_p]
lib.tllm_gemm_tactics_import.restype = ctypes.c_int32
lib.tllm_gemm_tactics_clear.restype = None
cfg = dict(bench.LLAMA_7B, num_layers=32)
dev = torch.device('cuda', 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
stream = torch.cuda.current_stream().cuda_stream
for mode in (sys.argv[1:] or ['sq', 'fp16', 'woq8']):
    int8_kv = mode != 'fp16'
    qm = bench.QM[mode] | (bench.INT8_KV if int8_kv else 0)
    w = bench.synth_weights(torch, cfg, mode, int8_kv, 1, 0, dev)
    lib.tllm_gemm_tactics_clear()
    s = NativeSession(dict(cfg, quant_mode=qm, tp_size=1, tp_rank=0))
    for k, t in w.items():
        s.set_tensor(k, t)
    s.finalize()
    for S in (128, 256, 512, 1024):
        ids = np.random.default_rng(1).integers(3, 32000, (1, S)).astype(np.int32)
        lens = np.array([S], np.int32)
        s.setup(1, S, 8)
        s.context(ids, lens, stream=stream)  # profiles the shapes of this M
        l_new = s.logits(stream=stream)
        need = lib.tllm_gemm_tactics_export(None, 0)
        buf = ctypes.create_string_buffer(need + 1)
        lib.tllm_gemm_tactics_export(buf, need + 1)
        table = buf.value.decode()
        mine = [e for e in table.split(';') if e and int(e.split(':')[1]) == S]
        old = re.sub(r'(\d+:\d+:\d+:\d+):65:', r'\1:64:', table)
        old = re.sub(r'(\d+:\d+:\d+:\d+):58:', r'\1:57:', old)
        res = {}
        for rnd in range(3):
            for name, t in (('profiled', table), ('65->64, 58->57', old)):
                lib.tllm_gemm_tactics_clear()
                assert lib.tllm_gemm_tactics_import(t.encode()) == 0
                s.context(ids, lens, stream=stream)
                if rnd == 0 and name != 'profiled':
                    print(f'   logits max |d| between the tables: {float(np.abs(s.logits(stream=stream) - l_new).max()):.3g}')
                ts = []
                for _ in range(3):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    s.context(ids, lens, stream=stream)
                    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
                res.setdefault(name, []).append(min(ts))
        lib.tllm_gemm_tactics_clear()
        lib.tllm_gemm_tactics_import(table.encode())
        print(f'{mode} S={S}: ' + ' | '.join(f'{n}: {min(v):.3f} ms' for n, v in res.items()) + '   table: ' + ' '.join(mine), flush=True)
    s.close()
    del w
    torch.cuda.empty_cache()
