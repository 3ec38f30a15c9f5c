"""SmoothQuant GEMM tile-shape sweep at the LLaMA-7B prefill shapes: every requested tllm_gemm_set_tile_cfg id is checked
exactly (against an int32 matmul of the same operands, same epilogue formula) on the first shape and then timed on all
four, interleaved round-robin inside one process (one box, one clock state).
    python tools/gemm_sweep.py [M] cfg [cfg ...]      e.g.  python tools/gemm_sweep.py 1024 6 8 20 63"""
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
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)


class GemmParams(ctypes.Structure):
    _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p),
                ('ldc', ctypes.c_int64)]


lib.tllm_gemm.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p]
lib.tllm_gemm.restype = ctypes.c_int32
lib.tllm_gemm_set_clock_probe.argtypes = [ctypes.c_void_p]
lib.tllm_gemm_set_clock_probe.restype = None
stream = torch.cuda.current_stream().cuda_stream
D, I = 4096, 11008
shapes = {'qkv': (3 * D, D), 'o_proj': (D, D), 'gate_or_up': (I, D), 'down': (D, I)}
if os.environ.get('SHAPES'):  # "N,K;N,K;..." instead of the four LLaMA-7B projections
    shapes = {f'{n}x{k}': (int(n), int(k)) for n, k in (x.split(',') for x in os.environ['SHAPES'].split(';'))}
torch.manual_seed(0)
res = {}
for name, (N, K) in shapes.items():
    PAD = int(os.environ.get('PAD', '0'))  # extra bytes per operand row (L2 channel experiments)
    a = torch.randint(-128, 128, (M, K + PAD), dtype=torch.int8, device=dev)[:, :K]
    w = torch.randint(-128, 128, (N, K + PAD), dtype=torch.int8, device=dev)[:, :K]
    sc = (torch.randint(1, 13, (N, ), device=dev).float() * 1e-4)
    sr = (torch.randint(1, 13, (M, ), device=dev).float() * 1e-3)
    c = torch.empty((M, N), dtype=torch.float16, device=dev)
    q = GemmParams(3, 1, M, N, K, a.data_ptr(), K + PAD, w.data_ptr(), K + PAD, sc.data_ptr(), sr.data_ptr(), 1, 1, c.data_ptr(), N)
    # exact reference: int32 accumulation in fp64 chunks (|sum| < 2^31 exactly representable), then the epilogue formula
    acc = torch.zeros((M, N), dtype=torch.float64, device=dev)
    for k0 in range(0, K, 2048):
        acc += a[:, k0:k0 + 2048].double() @ w[:, k0:k0 + 2048].double().t()
    ref = (acc.float() * (sc[None, :] * sr[:, None])).half()
    for cfg in cfgs:
        lib.tllm_gemm_set_tile_cfg(cfg)
        c.zero_()
        if lib.tllm_gemm(ctypes.byref(q), stream):
            raise RuntimeError(capi.last_error())
        torch.cuda.synchronize()
        bad = int((c != ref).sum().item())
        if bad and cfg < 21:  # 21.. are ablations (wrong on purpose)
            print(f'cfg {cfg} {name}: {bad} of {M * N} outputs differ from the exact reference')
        res.setdefault(cfg, {})[name] = {'bad': bad, 'us': []}
    # COLD=n: n copies of W and of X, a different one per launch (n x the bytes beyond the 256 MB Infinity Cache: every launch
    # finds its operands in HBM, as a layer of the prefill does)
    ncold = int(os.environ.get('COLD', '0'))
    wcopies = [w] + [w.clone() for _ in range(max(ncold - 1, 0))]
    acopies = [a] + [a.clone() for _ in range(max(ncold - 1, 0))]
    for rnd in range(5):
        for cfg in cfgs:
            lib.tllm_gemm_set_tile_cfg(cfg)
            lib.tllm_gemm(ctypes.byref(q), stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(20):
                if ncold:
                    q.w = wcopies[it % ncold].data_ptr()
                    q.a = acopies[it % ncold].data_ptr()
                lib.tllm_gemm(ctypes.byref(q), stream)
            e1.record()
            q.w, q.a = w.data_ptr(), a.data_ptr()
            torch.cuda.synchronize()
            res[cfg][name]['us'].append(e0.elapsed_time(e1) * 1e3 / 20)
            if os.environ.get('CLOCKS'):  # the shader clock the kernel held (s_memtime against the 100 MHz counter)
                dbg = torch.zeros(4096, dtype=torch.int64, device=dev)
                lib.tllm_gemm_set_clock_probe(ctypes.c_void_p(dbg.data_ptr()))
                for _ in range(8):
                    lib.tllm_gemm(ctypes.byref(q), stream)
                torch.cuda.synchronize()
                lib.tllm_gemm_set_clock_probe(None)
                d = dbg.view(-1, 2)[:128].double()
                mhz = (d[:, 0] / d[:, 1].clamp(min=1)).median().item() * 100.0
                res[cfg][name].setdefault('mhz', []).append(mhz)
lib.tllm_gemm_set_tile_cfg(0)
for cfg in cfgs:
    line = f'cfg {cfg:2d}'
    for name, (N, K) in shapes.items():
        r = res[cfg][name]
        us, med = min(r['us']), sorted(r['us'])[2]
        tops = 2.0 * M * N * K / us / 1e6
        line += f' | {name} {us:6.1f} us (med {med:6.1f}) {tops:5.0f} TOP/s {tops / 5000:.3f}{"" if not r["bad"] or cfg >= 21 else " WRONG"}'
        if r.get('mhz'):
            line += f' @ {sorted(r["mhz"])[len(r["mhz"]) // 2]:.0f} MHz'
    print(line)
