"""Fused SwiGLU SmoothQuant GEMM (gate | up of the 7B MLP, M rows): persistent form against the one-tile-per-workgroup form,
interleaved in one process.    python tools/dual_ab.py [M]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import torch  # noqa: E402

from tensorrt_llm.plugin import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N, K = 11008, 4096
lib = capi.load_library()
lib.tllm_gemm_set_tile_cfg.argtypes = [ctypes.c_int32]
lib.tllm_gemm_set_tile_cfg.restype = None


class GemmParams(ctypes.Structure):
    _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p),
                ('ldc', ctypes.c_int64)]


lib.tllm_gemm_swiglu_quant.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
lib.tllm_gemm_swiglu_quant.restype = ctypes.c_int32
dev = torch.device('cuda', 0)
torch.manual_seed(0)
a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
w1 = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
w2 = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
s1 = torch.randint(1, 13, (N, ), device=dev).float() * 2e-5
s2 = torch.randint(1, 13, (N, ), device=dev).float() * 2e-5
sr = torch.tensor([0.75], device=dev)
qs = torch.tensor([23.0], device=dev)
out = torch.empty((M, N), dtype=torch.int8, device=dev)
q = GemmParams(3, 2, M, N, K, a.data_ptr(), K, w1.data_ptr(), K, s1.data_ptr(), sr.data_ptr(), 1, 0, out.data_ptr(), N)
stream = torch.cuda.current_stream().cuda_stream


def run():
    if lib.tllm_gemm_swiglu_quant(ctypes.byref(q), w2.data_ptr(), s2.data_ptr(), qs.data_ptr(), stream):
        raise RuntimeError(capi.last_error())


res = {0: [], -2: []}
outs = {}
for rnd in range(5):
    for cfg in (0, -2):
        lib.tllm_gemm_set_tile_cfg(cfg)
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[cfg].append(e0.elapsed_time(e1) * 1e3 / 20)
        outs[cfg] = out.clone()
lib.tllm_gemm_set_tile_cfg(0)
for cfg, name in ((0, 'persistent'), (-2, 'one tile per workgroup')):
    us = min(res[cfg])
    print(f'M {M} gate|up + SwiGLU + quantiser, {name:24s}: {us:7.1f} us (median {sorted(res[cfg])[2]:7.1f})  '
          f'{4.0 * M * N * K / us / 1e6:6.0f} TOP/s = {4.0 * M * N * K / us / 1e6 / 5000:.3f} of 5 POP/s')
print('identical bytes:', bool(torch.equal(outs[0], outs[-2])))
