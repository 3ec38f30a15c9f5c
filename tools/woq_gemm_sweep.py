"""Weight-only prefill GEMM (gemm_woq.hip: dequantisation in the main loop) at the LLaMA-7B prefill shapes: every requested
tile id (101 = 256 x 192, 102 = 128 x 128, 103 = 256 x 192 two stages ahead, 104 = 256 x 192 on 4 waves, 0 = the launcher's rule)
timed interleaved in one process, plus the fp16 kernel on the same shapes.
    python tools/woq_gemm_sweep.py [M] [bits] cfg [cfg ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tensorrt_llm.plugin import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfgs = [int(x) for x in sys.argv[3:]] or [0]
lib = capi.load_library()
lib.tllm_gemm_set_tile_cfg.argtypes = [ctypes.c_int32]
lib.tllm_gemm_set_tile_cfg.restype = None


class GemmParams(ctypes.Structure):
    _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p), ('ldc', ctypes.c_int64)]


lib.tllm_gemm.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p]
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream().cuda_stream
D, I = 4096, 11008
shapes = {'qkv': (3 * D, D), 'o_proj': (D, D), 'gate_or_up': (I, D), 'down': (D, I)}
torch.manual_seed(0)
for name, (N, K) in shapes.items():
    a = torch.randn((M, K), dtype=torch.float16, device=dev)
    ldw = K if bits == 8 else K // 2
    w = torch.randint(0, 256, (N, ldw), dtype=torch.uint8, device=dev)
    sc = torch.full((N, ), 1e-2, dtype=torch.float16, device=dev)
    c = torch.empty((M, N), dtype=torch.float16, device=dev)
    wf = (torch.randn((N, K), device=dev) * 0.02).half()
    q = GemmParams(1 if bits == 8 else 2, 1, M, N, K, a.data_ptr(), K, w.data_ptr(), ldw, sc.data_ptr(), None, 0, 0, c.data_ptr(), N)
    qf = GemmParams(0, 1, M, N, K, a.data_ptr(), K, wf.data_ptr(), 2 * K, None, None, 0, 0, c.data_ptr(), N)
    res = {}
    for rnd in range(3):
        for cfg in cfgs + ['fp16']:
            par = qf if cfg == 'fp16' else q
            lib.tllm_gemm_set_tile_cfg(0 if cfg == 'fp16' else cfg)
            for _ in range(2):
                assert lib.tllm_gemm(ctypes.byref(par), stream) == 0, capi.last_error()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.tllm_gemm(ctypes.byref(par), stream)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(cfg, []).append(e0.elapsed_time(e1) * 100)
    lib.tllm_gemm_set_tile_cfg(0)
    print(f'{name:11s} M={M} N={N} K={K} int{bits}: ' + ' | '.join(
        f'{cfg}: {min(v):6.1f} us {2.0 * M * N * K / min(v) / 1e6:5.0f} TF/s' for cfg, v in res.items()))
