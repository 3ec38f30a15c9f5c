"""fp16 prefill GEMM at the LLaMA-7B prefill shapes: every requested kernel id (gemm_glds.hip lock-step ids 1..12, gemm_sqp.hip
phased ids 50..53; 0 = the launcher's own choice) checked against an fp64-accumulated product of the same operands and timed
interleaved in one process.
    python tools/fp16_gemm_sweep.py [M] cfg [cfg ...]      e.g.  python tools/fp16_gemm_sweep.py 1024 6 8 50 51 52 53"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
import torch  # noqa: E402

from tensorrt_llm.plugin import capi  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfgs = [int(x) for x in sys.argv[2:]] or [0]
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
if os.environ.get('SHAPES'):
    shapes = {f'{n}x{k}': (int(n), int(k)) for n, k in (x.split(',') for x in os.environ['SHAPES'].split(';'))}
torch.manual_seed(0)
for name, (N, K) in shapes.items():
    a = torch.randn((M, K), dtype=torch.float16, device=dev)
    w = (torch.randn((N, K), device=dev) * 0.02).half()
    c = torch.empty((M, N), dtype=torch.float16, device=dev)
    q = GemmParams(0, 1, M, N, K, a.data_ptr(), K, w.data_ptr(), 2 * K, None, None, 0, 0, c.data_ptr(), N)
    ref = (a.double() @ w.double().t())
    res, err = {}, {}
    for cfg in cfgs:
        lib.tllm_gemm_set_tile_cfg(cfg)
        c.zero_()
        assert lib.tllm_gemm(ctypes.byref(q), stream) == 0, capi.last_error()
        torch.cuda.synchronize()
        d = (c.double() - ref).abs()
        err[cfg] = (float(d.max()), float((d / (ref.abs() + 1e-2)).max()))
    for rnd in range(5):
        for cfg in cfgs:
            lib.tllm_gemm_set_tile_cfg(cfg)
            for _ in range(2):
                lib.tllm_gemm(ctypes.byref(q), stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                lib.tllm_gemm(ctypes.byref(q), stream)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(cfg, []).append(e0.elapsed_time(e1) * 1e3 / 20)
    lib.tllm_gemm_set_tile_cfg(0)
    print(f'{name:11s} M={M} N={N} K={K} fp16: ' + ' | '.join(
        f'{cfg}: {min(v):6.1f} us (med {sorted(v)[2]:6.1f}) {2.0 * M * N * K / min(v) / 1e6:5.0f} TF/s {2.0 * M * N * K / min(v) / 1e6 / 2500:.3f} '
        f'[max err {err[cfg][0]:.3g}]' for cfg, v in res.items()))
