#!/usr/bin/env python3
"""Headline benchmark: decode tokens/sec of LLaMA-7B, batch 1, 1024-token context (BASELINE.json `metric`),
int8 (SmoothQuant per-channel int8 weights+activations, int8 KV cache — BASELINE.json configs[3]) on N GPUs of
one node with tensor parallelism TP = N (the reference's only sharding).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config sq|woq8|woq4|fp16] [--context 1024]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one generation step (one new token for the whole batch) through all 32 layers + lm_head + greedy
sampler, replayed from the captured hipGraph with the KV cache holding `context` tokens (+ the tokens generated so
far).  Weights are synthetic (seeded random of the LLaMA-7B architecture, generated on the GPU in the storage format
of the chosen config); inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line with
`roofline` (the launch that takes most of the step, picked at run time as max(calls x average duration) over the launches the
step is made of - all of them are listed in `roofline.kernels`; HBM-bound) and `cpu_baseline` (HF transformers LLaMA on the host
CPU, the reference's run_hf.py path, bounded sample) objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
sys.path.insert(0, ROOT)

LLAMA_7B = dict(num_layers=32, num_heads=32, hidden_size=4096, inter_size=11008, vocab_size=32000,
                max_position_embeddings=2048, rms_norm_eps=1e-6)  # T/examples/llama_quant/build.py:59-71
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
QM = dict(fp16=0, woq8=2, woq4=1, sq=2 | 4 | 8)  # QuantMode bits (T/tensorrt_llm/quantization/mode.py:6-21)
INT8_KV = 32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=128)
    ap.add_argument('--warmup', type=int, default=8)
    ap.add_argument('--config', default='sq', choices=['sq', 'woq8', 'woq4', 'fp16'])
    ap.add_argument('--context', type=int, default=1024)
    ap.add_argument('--layers', type=int, default=32, help='debug only: fewer layers (the result line says so)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--two-launch-attention', action='store_true',
                    help='A/B: QKV projection and attention as two launches (session key fuse_qkv_attention = 0)')
    ap.add_argument('--gemv-o-projection', action='store_true',
                    help='A/B: the O-projection as a GEMV launch of its own (session key fuse_o_projection = 0)')
    ap.add_argument('--one-launch-mlp', action='store_true',
                    help='A/B: the gated MLP of a decode step as ONE launch (session key fuse_mlp=1; measured slower, off by default)')
    ap.add_argument('--session-key', action='append', default=[], metavar='KEY=INT', help='debug: extra session key(s) for A/B runs')
    ap.add_argument('--no-prefill', dest='prefill', action='store_false', help='skip the prefill (TTFT) and SQ-GEMM MFMA reports')
    ap.add_argument('--no-fp16-ref', action='store_true', help='skip the fp16 config run used for the int8/fp16 ratio')
    ap.add_argument('--no-batch-sweep', dest='batch_sweep', action='store_false',
                    help='skip the side report of decode tokens/s at 4 and 8 sequences (N = 1 only; not part of the metric)')
    ap.add_argument('--no-tp-prediction', dest='tp_prediction', action='store_false',
                    help='skip the side report of one rank\'s step time at the tp = 2 / 4 / 8 extents without collectives')
    ap.add_argument('--no-parity', dest='parity', action='store_false',
                    help='skip the 7B accuracy report (GPU engines vs HF fp32 on the host CPU on identical weights)')
    ap.add_argument('--parity-new-tokens', type=int, default=128, help='new tokens per prompt of the accuracy report (SURVEY 8d: 128)')
    ap.add_argument('--parity-prompts', type=int, default=8, help='seeded prompts of the accuracy report')
    ap.add_argument('--cpu-tokens', type=int, default=0, help='CPU-baseline decode steps (0 = sized to ~20 s)')
    ap.add_argument('--cpu-threads', type=int, default=0)
    ap.add_argument('--cpu-sweep', default='', help='debug: comma list of thread counts to try (stderr)')
    return ap.parse_args()


def synth_weights(torch, cfg, mode, int8_kv, tp, rank, dev):
    """Seeded random LLaMA-7B-shaped weights directly in the engine's storage formats (per-rank shards)."""
    D, I, V, H = cfg['hidden_size'], cfg['inter_size'], cfg['vocab_size'], cfg['num_heads']
    Dr, Ir, Vr = D // tp, I // tp, (V + tp - 1) // tp
    g_rep = torch.Generator(device=dev).manual_seed(0)  # replicated tensors: same on every rank
    g = torch.Generator(device=dev).manual_seed(1000 + rank)  # shards
    t = {}

    def u(gen, shape, r):  # fp16 U(-r, r)
        return ((torch.rand(shape, generator=gen, device=dev, dtype=torch.float32) * 2 - 1) * r).half()

    def xavier(n, k):
        return (6.0 / (n + k)) ** 0.5

    t['vocab_embedding.weight'] = (torch.randn((V, D), generator=g_rep, device=dev) * 0.5).half()
    t['ln_f.weight'] = (1 + 0.1 * (torch.rand(D, generator=g_rep, device=dev) * 2 - 1)).half()
    t['lm_head.weight'] = u(g, (Vr, D), xavier(V, D))
    f32 = lambda v: torch.tensor([v], dtype=torch.float32, device=dev)

    def linear(prefix, n, k, fan):
        r = fan
        if mode == 'fp16':
            t[prefix + '.weight'] = u(g, (n, k), r)
        elif mode == 'sq':
            t[prefix + '.weight'] = torch.randint(-127, 128, (n, k), generator=g, device=dev, dtype=torch.int8)
            # static per-channel SmoothQuant: y = acc * per_channel_scale[n] * act_scale, activations ~ 28 / unit
            t[prefix + '.per_channel_scale'] = torch.full((1, n), r / 127.0 / 28.0, dtype=torch.float32, device=dev)
            t[prefix + '.act_scale'] = torch.ones((1, 1), dtype=torch.float32, device=dev)
        else:
            bits = 8 if mode == 'woq8' else 4
            row = k if bits == 8 else k // 2
            t[prefix + '.weight'] = torch.randint(0, 256, (n, row), generator=g, device=dev, dtype=torch.uint8)
            t[prefix + '.per_channel_scale'] = torch.full((n, ), r / (127.0 if bits == 8 else 7.0), dtype=torch.float16,
                                                          device=dev)

    for i in range(cfg['num_layers']):
        p = f'layers.{i}.'
        t[p + 'input_layernorm.weight'] = (1 + 0.1 * (torch.rand(D, generator=g_rep, device=dev) * 2 - 1)).half()
        t[p + 'post_layernorm.weight'] = (1 + 0.1 * (torch.rand(D, generator=g_rep, device=dev) * 2 - 1)).half()
        linear(p + 'attention.qkv', 3 * Dr, D, xavier(3 * D, D))
        linear(p + 'attention.dense', D, Dr, xavier(D, D))
        linear(p + 'mlp.fc', Ir, D, xavier(I, D))
        linear(p + 'mlp.gate', Ir, D, xavier(I, D))
        linear(p + 'mlp.proj', D, Ir, xavier(D, I))
        if mode == 'sq':
            t[p + 'input_layernorm.scale_to_int'] = f32(28.0)
            t[p + 'post_layernorm.scale_to_int'] = f32(28.0)
            t[p + 'attention.quantization_scaling_factor'] = f32(60.0)
            t[p + 'mlp.quantization_scaling_factor'] = f32(40.0)
        if int8_kv:
            t[p + 'attention.kv_orig_quant_scale'] = f32(127.0 / 6.0)
            t[p + 'attention.kv_quant_orig_scale'] = f32(6.0 / 127.0)
    return t


def _fused_nit(smax, int8_kv):
    """cache rows per lane group of the one-launch projection + attention (qkv_attn_fused.hip pick_nit): names the instance"""
    ngrp = 8 * (8 if int8_kv else 4)
    need = -(-smax // (8 * ngrp))
    for n in (1, 2, 3, 4, 6, 8):
        if need <= n:
            return n
    return 0


def step_launches(cfg, mode, world, form, l_mean, int8_kv, smax):
    """The launches one generation step is made of, as this session runs it (tllm_session_decode_form), each with its kernel
    name as rocprofv3 prints it, its launches per step and its ALGORITHMIC HBM bytes per launch (weights +
    per-channel scales + KV read at the mean context of the timed steps + the activation rows in and out)."""
    D, H, L = cfg['hidden_size'], cfg['num_heads'], cfg['num_layers']
    Dh = D // H
    Dr, Ir, Hr = D // world, cfg['inter_size'] // world, H // world
    Vr = (cfg['vocab_size'] + world - 1) // world
    wb = {'sq': 1.0, 'woq8': 1.0, 'woq4': 0.5, 'fp16': 2.0}[mode]
    sb = {'sq': 4, 'woq8': 2, 'woq4': 2, 'fp16': 0}[mode]  # per-output-channel scale
    wt = {'sq': 3, 'woq8': 1, 'woq4': 2, 'fp16': 0}[mode]
    e = 1 if int8_kv else 2
    row = D * 2  # one fp16 activation row
    qkv = 3 * Dr * D * wb + 3 * Dr * sb + 2 * row + 3 * Dr * 2
    kv = 2 * Hr * Dh * l_mean * e + 2 * Hr * Dh * e
    o = D * Dr * wb + D * sb + Dr * (1 if mode == 'sq' else 2) + 2 * row
    gate_up = 2 * Ir * D * wb + 2 * Ir * sb + 2 * row + Ir * (1 if mode == 'sq' else 2)
    down = D * Ir * wb + D * sb + Ir * (1 if mode == 'sq' else 2) + 2 * row
    head = Vr * D * 2 + 2 * row + Vr * 4
    out = []
    if form & 1:
        nit = _fused_nit(smax, int8_kv)
        name = 'qkv_attn_fused_kernel<%d, %s, %d>' % (nit, 'true' if int8_kv else 'false', {'sq': 0, 'woq8': 1, 'fp16': 2, 'woq4': 3}.get(mode, 0))
        b, what = qkv + kv - 3 * Dr * 2, 'RMSNorm -> QKV GEMV -> RoPE -> cache append -> attention'
        if form & 2:
            b, what = b + o - Dr * (1 if mode == 'sq' else 2), what + ' -> O-projection + residual'
        out.append(dict(key='front', id=7, kernel=name, what=what, bytes=b, calls=L, match=['qkv_attn_fused_kernel']))
    else:
        out.append(dict(key='qkv', id=1, kernel='gemv_kernel<%d, 1, 0, 1, 2, 4>' % wt, what='RMSNorm -> QKV GEMV', bytes=qkv, calls=L,
                        match=['gemv_kernel<%d, 1, 0' % wt]))
        out.append(dict(key='attention', id=2, kernel='mmha_partial_kernel', what='RoPE -> cache append -> split attention + merge',
                        bytes=kv + 3 * Dr * 2, calls=L, match=['mmha_']))
    if not form & 2:
        out.append(dict(key='o_proj', id=4, kernel='gemv (O-projection + residual)', what='O-projection + residual', bytes=o, calls=L,
                        match=['gemv_ksplit_kernel', 'gemv_kernel<%d, 0, 0' % wt, 'gemv_kernel<%d, 2, 0' % wt]))
    if form & 4:
        # (one launch: the intermediate row is written once and read once per workgroup, 256 x Ir bytes of L2 / fabric traffic that
        #  is not part of the algorithmic HBM bytes)
        out.append(dict(key='mlp', id=8, kernel='mlp_fused_kernel<%d>' % ((Ir + 1023) // 1024),
                        what='RMSNorm -> gate|up GEMV -> SwiGLU -> down projection + residual', bytes=gate_up + down - 2 * row, calls=L,
                        match=['mlp_fused_kernel']))
    else:
        out.append(dict(key='gate_up', id=5, kernel='gemv_kernel<%d, 1, 1, 1, 2, 4>' % wt, what='RMSNorm -> gate|up GEMV -> SwiGLU',
                        bytes=gate_up, calls=L, match=['gemv_kernel<%d, 1, 1' % wt]))
        out.append(dict(key='down', id=6, kernel='gemv_ksplit_kernel<3, 3>' if mode == 'sq' else 'gemv (down projection + residual)',
                        what='down projection + residual', bytes=down, calls=L,
                        match=['gemv_ksplit_kernel', 'gemv_kernel<%d, 0, 0' % wt, 'gemv_kernel<%d, 2, 0' % wt]))
    out.append(dict(key='head', id=None, kernel='gemv_kernel<0, 1, 0, ...> (ln_f -> lm_head, fp32 logits)', what='final RMSNorm -> lm_head GEMV',
                    bytes=head, calls=1, match=['gemv_kernel<0, 1, 0']))
    return out


def pmc_traffic(launch, rows):
    """HBM bytes per launch of this kernel from the committed rocprofv3 --pmc summary (FETCH_SIZE x 2 + WRITE_SIZE): among the
    rows whose name matches, the one whose volume is nearest to the algorithmic bytes (several shapes share a template instance)."""
    import math
    best = None
    for r in rows:
        if not any(m in r['kernel'] for m in launch['match']):
            continue
        t = r['fetch_bytes'] + r['write_bytes']
        if t <= 0:
            continue
        d = abs(math.log(t / launch['bytes']))
        if d < math.log(1.6) and (best is None or d < best[0]):
            best = (d, t, r['kernel'])
    return (best[1], best[2]) if best else (None, None)


def run_config(torch, dist, args, mode, rank, world, dev):
    from tensorrt_llm.runtime.native import NativeSession
    cfg = dict(LLAMA_7B, num_layers=args.layers)
    int8_kv = mode != 'fp16'  # BASELINE.json configs: fp16 + fp16 KV; every int8 config with int8 KV
    qm = QM[mode] | (INT8_KV if int8_kv else 0)
    sess = NativeSession(dict(cfg, quant_mode=qm, tp_size=world, tp_rank=rank, fuse_qkv_attention=0 if args.two_launch_attention else -1,
                              fuse_o_projection=0 if getattr(args, 'gemv_o_projection', False) else -1,
                              fuse_mlp=1 if getattr(args, 'one_launch_mlp', False) else 0,
                              **{k: int(v) for k, v in (kv.split('=') for kv in getattr(args, 'session_key', []))}))
    weights = synth_weights(torch, cfg, mode, int8_kv, world, rank, dev)
    for k, v in weights.items():
        sess.set_tensor(k, v)
    sess.finalize()
    K, W = args.steps, args.warmup
    sess.setup(1, args.context, 2 * K + W + 4)  # the timed K steps + the same K once more under an event pair (below)
    stream = torch.cuda.current_stream().cuda_stream
    sess.fake_context(args.context, seed=1, stream=stream)
    sess.step(2, use_graph=False, stream=stream)  # eager: lazy init outside the capture
    # captures the graph + W untimed warm-up steps.  Insurance for N > 1: if the capture fails (e.g. a collective that cannot
    # be captured on this software stack) the run continues with eager launches instead of dying - the result line says so.
    use_graph = True
    try:
        sess.step(max(W, 1), use_graph=True, stream=stream)
    except RuntimeError as e:
        print(f'[bench rank {rank}] graph capture failed ({e}); falling back to eager steps', file=sys.stderr, flush=True)
        use_graph = False
        sess.step(max(W, 1), use_graph=False, stream=stream)  # the same collectives the other ranks' warm-up issues
    torch.cuda.synchronize()
    if world > 1:
        flag = torch.tensor([1 if use_graph else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        use_graph = bool(flag.item())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # The timed region holds NOTHING but the K steps (the device-time event pair has a pass of its own below).
    t0 = time.perf_counter()
    sess.step(K, use_graph=use_graph, stream=stream)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # device time of the same K steps between a HIP event pair on the same stream, in a pass of its own
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sess.step(K, use_graph=use_graph, stream=stream)
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1)
    if world > 1:
        tt = torch.tensor([wall], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall = float(tt.item())
    logits = sess.logits(stream=stream)
    finite = bool((logits == logits).all() and abs(logits).max() < 1e30)
    # context length seen by the timed steps: context + 2 + W ... + K (mean)
    l_mean = args.context + 2 + max(W, 1) + (K - 1) / 2.0
    step_bytes = sess.step_bytes(int(round(l_mean)))
    res = dict(graph=use_graph, mode=mode, wall_s=wall, dev_ms=dev_ms, ms_per_step=wall * 1e3 / K, tokens_per_s=K / wall, finite=finite,
               step_bytes=step_bytes, mean_context=l_mean)
    # instrumented pass (eager, an event pair around every launch): the head GEMV's and the sampler's time, the collectives
    prof = sess.profile(8, stream=stream)
    res['profile'] = prof
    # live timing of every launch the step is made of (one HIP event pair around 4 x 32 back-to-back launches, every launch on
    # its own layer's weights = cold HBM as in a real step) next to its algorithmic HBM bytes
    form = sess.decode_form()
    res['decode_form'] = form
    res['launches'] = step_launches(cfg, mode, world, form, l_mean, int8_kv, args.context + 2 * K + W + 4)
    for ln in res['launches']:
        if ln['id'] is not None:
            ln['avg_us'] = sess.time_kernel(ln['key'], 4, stream=stream)[0]
        else:  # the head GEMV: one launch per step, from the instrumented pass
            ln['avg_us'] = prof['gemv_head'][0] * 1e3 / max(prof['gemv_head'][1], 1)
    res['kernel_us'] = {ln['key']: ln['avg_us'] for ln in res['launches']}
    # time-to-first-token: the real context phase (MFMA GEMMs + flash attention + KV write) on a random 1024-token
    # prompt - reported beside the decode metric, not part of it
    if args.prefill and mode == args.config:
        import numpy as np
        ids = np.random.default_rng(1).integers(3, cfg['vocab_size'], (1, args.context)).astype(np.int32)
        lens = np.array([args.context], np.int32)
        sess.context(ids, lens, stream=stream)  # warm-up
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            sess.context(ids, lens, stream=stream)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        res['prefill_ms'] = min(ts) * 1e3
    # side report (not the metric, which is batch 1): the same step with 4 and 8 sequences - the weights are read once per step
    # whatever the batch, so this is what a serving deployment of the path gets per GPU (profiles/r04_batch_sweep.txt)
    if getattr(args, 'batch_sweep', False) and mode == args.config and world == 1:
        res['batch'] = {}
        try:
            for B in (4, 8):
                kb = 32
                sess.setup(B, args.context, 2 * kb + 16)
                sess.fake_context(args.context, seed=1, stream=stream)
                sess.step(2, use_graph=False, stream=stream)
                sess.step(4, use_graph=True, stream=stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sess.step(kb, use_graph=True, stream=stream)
                torch.cuda.synchronize()
                dtb = time.perf_counter() - t1
                res['batch'][str(B)] = {'ms_per_step': dtb * 1e3 / kb, 'tokens_per_s': B * kb / dtb}
        except Exception as e:  # the decode metric must not depend on the side report
            res['batch']['error'] = repr(e)
    sess.close()
    del weights
    torch.cuda.empty_cache()
    return res


def tp_rank_prediction(torch, args, dev):
    """Side report for the first multi-GPU run to be held against: ONE rank's generation step at the tensor-
    parallel extents tp = 2 / 4 / 8 (heads, FFN columns, vocabulary / tp), timed on this one GPU WITHOUT its
    collectives (session key no_comm: the all-reduces and the logits all-gather are skipped, every other launch is the rank's own).
    tokens/s of the tp-way job = 1 / (this + the 64 all-reduces + the all-gather of a step); `hbm_floor_ms` = the rank's bytes at
    6.3 TB/s.  Timing only - the hidden states are one rank's partial sums."""
    from tensorrt_llm.runtime.native import NativeSession
    out = {'note': 'one rank, no collectives (session key no_comm = 1), graph replay, this GPU; timing only'}
    cfg = dict(LLAMA_7B, num_layers=args.layers)
    qm = QM['sq'] | INT8_KV
    stream = torch.cuda.current_stream().cuda_stream
    for tp in (2, 4, 8):
        try:
            sess = NativeSession(dict(cfg, quant_mode=qm, tp_size=tp, tp_rank=0, no_comm=1))
            w = synth_weights(torch, cfg, 'sq', True, tp, 0, dev)
            for k, v in w.items():
                sess.set_tensor(k, v)
            sess.finalize()
            K = 64
            sess.setup(1, args.context, K + 16)
            sess.fake_context(args.context, seed=1, stream=stream)
            sess.step(2, use_graph=False, stream=stream)
            sess.step(4, use_graph=True, stream=stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sess.step(K, use_graph=True, stream=stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            nbytes = sess.step_bytes(args.context + 8 + K // 2)
            out[f'tp{tp}'] = {'rank_ms_per_step_without_comm': dt * 1e3 / K, 'rank_hbm_bytes_per_step': nbytes,
                              'hbm_floor_ms': nbytes / 6.3e12 * 1e3, 'tokens_per_s_if_comm_were_free': K / dt,
                              'collectives_per_step': 2 * args.layers + 1}
            sess.close()
            del w
            torch.cuda.empty_cache()
        except Exception as e:  # the decode metric must not depend on the side report
            out[f'tp{tp}'] = {'error': repr(e)}
    return out


def sq_gemm_mfma_report(torch, dev, M=1024):
    """BASELINE.json's second target: the SmoothQuant GEMM at the prefill shapes (M = 1024) as
    a fraction of the dense int8 MFMA peak (5 POP/s).  Random int8 operands (constant operands clock ~19 % higher and
    flatter the number), 20 launches per shape timed with one event pair on torch's current stream."""
    import ctypes

    from tensorrt_llm.plugin import capi
    lib = capi.load_library()

    class GemmParams(ctypes.Structure):
        _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                    ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                    ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                    ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p),
                    ('ldc', ctypes.c_int64)]

    lib.tllm_gemm.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p]
    lib.tllm_gemm.restype = ctypes.c_int32
    D, I = LLAMA_7B['hidden_size'], LLAMA_7B['inter_size']
    shapes = {'qkv': (3 * D, D), 'o_proj': (D, D), 'gate_or_up': (I, D), 'down': (D, I)}
    out = {}
    stream = torch.cuda.current_stream().cuda_stream
    for name, (N, K) in shapes.items():
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
        w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
        sc = torch.full((N, ), 1e-3, dtype=torch.float32, device=dev)
        sr = torch.full((M, ), 1e-2, dtype=torch.float32, device=dev)
        c = torch.empty((M, N), dtype=torch.float16, device=dev)
        q = GemmParams(3, 1, M, N, K, a.data_ptr(), K, w.data_ptr(), K, sc.data_ptr(), sr.data_ptr(), 1, 1, c.data_ptr(), N)
        # the kernel is the one the on-device tactic profile finds fastest for this shape on THIS box (what a session does at
        # setup, tllm_gemm_profile; reference: int8_gemm_template.h:372-457) - the static rule's pick is reported beside it
        lib.tllm_gemm_tactics_clear()
        tactic, tactic_us = ctypes.c_int32(0), ctypes.c_float(0)
        lib.tllm_gemm_profile.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        if lib.tllm_gemm_profile(3, M, N, K, ctypes.byref(tactic), ctypes.byref(tactic_us), stream):
            raise RuntimeError(capi.last_error())
        for _ in range(3):
            if lib.tllm_gemm(ctypes.byref(q), stream):
                raise RuntimeError(capi.last_error())
        iters, reps = 20, []
        for _ in range(5):  # clocks move with load and temperature: report the best and the median of 5 x 20 launches
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                lib.tllm_gemm(ctypes.byref(q), stream)
            e1.record()
            torch.cuda.synchronize()
            reps.append(e0.elapsed_time(e1) * 1e3 / iters)
        us, us_med = min(reps), sorted(reps)[len(reps) // 2]
        # the static "fewest workgroup rounds" rule on the same operands, same warm state (usually the same kernel at M = 1024)
        static_us = None
        try:
            lib.tllm_gemm_tactics_clear()
            for _ in range(3):
                lib.tllm_gemm(ctypes.byref(q), stream)
            sreps = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    lib.tllm_gemm(ctypes.byref(q), stream)
                e1.record()
                torch.cuda.synchronize()
                sreps.append(e0.elapsed_time(e1) * 1e3 / iters)
            static_us = min(sreps)
        except Exception:
            pass
        if tactic.value > 0:  # back to the profiled choice for the clock probe below
            lib.tllm_gemm_tactics_import(f'3:{M}:{N}:{K}:{int(tactic.value)}:{float(tactic_us.value):.2f};'.encode())
        tops = 2.0 * M * N * K / us / 1e6
        out[name] = {'M': M, 'N': N, 'K': K, 'us': us, 'us_median': us_med, 'TOP/s': tops, 'frac_of_5POPs': tops / 5000.0,
                     'tactic': int(tactic.value), 'tactic_profile_us': float(tactic_us.value), 'static_rule_us': static_us}
        # what the SAME pipeline does with every memory operation removed (ablation id 33 of the 256 x 192 phased kernel: MFMAs + epilogue
        # only, wrong results on purpose): the ceiling of this kernel structure at this M - one round of workgroups, epilogue exposed
        if name == 'qkv' and M == 1024:
            try:
                lib.tllm_gemm_set_tile_cfg.argtypes = [ctypes.c_int32]
                lib.tllm_gemm_set_tile_cfg.restype = None
                lib.tllm_gemm_set_tile_cfg(33)
                for _ in range(3):
                    lib.tllm_gemm(ctypes.byref(q), stream)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters):
                    lib.tllm_gemm(ctypes.byref(q), stream)
                e1.record()
                torch.cuda.synchronize()
                mo = e0.elapsed_time(e1) * 1e3 / iters
                out[name]['mfma_only_us'] = mo
                out[name]['frac_of_the_mfma_only_form'] = mo / us
            except Exception as e:
                print(f'[bench] mfma-only ablation failed: {e!r}', file=sys.stderr)
            finally:
                lib.tllm_gemm_set_tile_cfg(0)
                if tactic.value > 0:
                    lib.tllm_gemm_tactics_import(f'3:{M}:{N}:{K}:{int(tactic.value)}:{float(tactic_us.value):.2f};'.encode())
        # the clock the chip held under this kernel (it clocks to its power budget: dense random-operand int8 MFMA work next to
        # the LDS / L2 traffic that feeds it runs well below 2.4 GHz): every workgroup reports its shader cycles against the
        # constant 100 MHz counter (tllm_gemm_set_clock_probe); 5 POP/s is the nominal peak AT 2.4 GHz
        try:
            probe = torch.zeros(8192, dtype=torch.int64, device=dev)
            lib.tllm_gemm_set_clock_probe.argtypes = [ctypes.c_void_p]
            lib.tllm_gemm_set_clock_probe.restype = None
            lib.tllm_gemm_set_clock_probe(ctypes.c_void_p(probe.data_ptr()))
            for _ in range(8):
                lib.tllm_gemm(ctypes.byref(q), stream)
            torch.cuda.synchronize()
            lib.tllm_gemm_set_clock_probe(None)
            d = probe.view(-1, 2)[:128].double()
            ok = d[:, 1] > 0
            if bool(ok.any()):
                mhz = float((d[ok, 0] / d[ok, 1]).median().item() * 100.0)
                out[name]['shader_MHz_held'] = mhz
                out[name]['frac_of_peak_at_held_clock'] = tops / (5000.0 * mhz / 2400.0)
        except Exception as e:  # side report only
            out[name]['shader_MHz_held'] = None
            print(f'[bench] clock probe failed: {e!r}', file=sys.stderr)
    return out


def _default_cpu_threads():
    """Threads for the CPU baseline: the physical cores of the host, capped at 64 - a batch-1 decode step is a chain
    of memory-bound GEMVs and oversubscribing a large multi-socket host makes it slower, not faster (measured on the
    256-thread GPU host: DESIGN.md, Measurement)."""
    n = os.cpu_count() or 1
    try:
        import psutil
        n = psutil.cpu_count(logical=False) or n
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(n_tokens, context, threads=0, sweep=None, model=None, layers=32):
    """HF transformers LlamaForCausalLM on the host CPU (the reference's run_hf.py flow,
    T/examples/llama_quant/run_hf.py:41-104, minus .cuda()): greedy decode of `n_tokens` tokens at batch 1 with a
    `context`-token KV cache.  Bounded sample: synthetic KV cache instead of a CPU prefill (13 TFLOP), a handful of tokens.
    `model`: the fp32 CPU model of the parity run (the seeded 7B parent); without it, weights tiled from a random pool."""
    import torch
    try:
        import transformers
        from transformers import LlamaConfig, LlamaForCausalLM
        from transformers.cache_utils import DynamicCache
    except Exception as e:  # pragma: no cover
        return dict(value=None, unit='tokens/s', cores=os.cpu_count(), kind='reference', sample=f'transformers unavailable: {e}')
    cores = threads or _default_cpu_threads()
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:
        avail = 0
    dtype = torch.float32 if (avail > 80e9 or model is not None) else torch.bfloat16
    torch.set_num_threads(cores)
    cfg = LlamaConfig(hidden_size=4096, num_attention_heads=32, num_key_value_heads=32, intermediate_size=11008,
                      vocab_size=32000, num_hidden_layers=layers, max_position_embeddings=2048, rms_norm_eps=1e-6,
                      attention_bias=False, tie_word_embeddings=False)
    t_build = time.perf_counter()
    weights = 'the seeded fp16 parent of the parity run (identical to the GPU engines\' weights)'
    if model is None:
        weights = 'synthetic weights tiled from a random pool'
        with torch.device('meta'):
            model = LlamaForCausalLM(cfg)
        model = model.to_empty(device='cpu').to(dtype).eval()
        pool = (torch.rand(1 << 24) * 2 - 1).mul_(0.02).to(dtype)
        with torch.no_grad():
            for name, p in model.named_parameters():
                flat = p.data.view(-1)
                if 'norm' in name:
                    flat.fill_(1.0)
                    continue
                n = flat.numel()
                for off in range(0, n, pool.numel()):
                    m = min(pool.numel(), n - off)
                    flat[off:off + m].copy_(pool[:m])
            # rotary buffers are not parameters: re-create them
            if hasattr(model.model, 'rotary_emb'):
                model.model.rotary_emb = type(model.model.rotary_emb)(config=cfg)
    cache = DynamicCache(config=cfg) if 'config' in DynamicCache.__init__.__code__.co_varnames else DynamicCache()
    kv = (torch.rand(1, 32, context, 128) * 2 - 1).to(dtype)
    for li in range(layers):
        cache.update(kv.clone(), kv.clone(), li)
    build_s = time.perf_counter() - t_build
    ids = torch.tensor([[3]])
    pos = context
    with torch.no_grad():
        # one untimed token (page-in), then the timed ones
        out = model(input_ids=ids, past_key_values=cache, use_cache=True, position_ids=torch.tensor([[pos]]))
        ids = out.logits[:, -1].argmax(-1, keepdim=True)
        pos += 1
        for th in (sweep or []):  # debug: tokens/s at other thread counts (stderr only)
            torch.set_num_threads(th)
            t0 = time.perf_counter()
            out = model(input_ids=ids, past_key_values=cache, use_cache=True, position_ids=torch.tensor([[pos]]))
            pos += 1
            print(f'[cpu_baseline sweep] threads={th}: {1.0 / (time.perf_counter() - t0):.3f} tok/s', file=sys.stderr)
        torch.set_num_threads(cores)
        if n_tokens <= 0:  # size the sample to ~20 s of CPU work from one probe token
            t0 = time.perf_counter()
            out = model(input_ids=ids, past_key_values=cache, use_cache=True, position_ids=torch.tensor([[pos]]))
            ids = out.logits[:, -1].argmax(-1, keepdim=True)
            pos += 1
            n_tokens = int(max(2, min(64, round(20.0 / max(time.perf_counter() - t0, 1e-3)))))
        t0 = time.perf_counter()
        for _ in range(n_tokens):
            out = model(input_ids=ids, past_key_values=cache, use_cache=True, position_ids=torch.tensor([[pos]]))
            ids = out.logits[:, -1].argmax(-1, keepdim=True)
            pos += 1
        dt = time.perf_counter() - t0
    return dict(value=n_tokens / dt, unit='tokens/s', cores=cores, kind='reference',
                sample=(f'HF transformers {transformers.__version__} LlamaForCausalLM (reference run_hf.py path) on the host CPU, '
                        f'{str(dtype).split(".")[-1]}, {cores} threads, LLaMA-7B ({layers} layers), {weights}, batch 1, '
                        f'{n_tokens} greedy decode steps at context {context} (synthetic KV cache, no prefill), '
                        f'{dt:.1f} s timed, {build_s:.0f} s untimed set-up'))


def main():
    args = parse()
    import torch
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP path has no CPU fallback)')
    if args.gpus > 1 and 'RANK' not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher - one process per GPU under torch.distributed.run
        # (the reference's model is the same, one process per GPU under mpirun: PY/_utils.py:181-190, Q/run.py:83-93)
        shared = os.environ.get('TLLM_TEST_SHARED_GPU') == '1'
        if torch.cuda.device_count() < args.gpus and not shared:
            raise SystemExit(f'bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible')
        import socket
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
        sock.close()
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL / hipIpc between the rank processes need it
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print('[bench] re-launching as: ' + ' '.join(cmd), file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f'bench.py --gpus {args.gpus} inside a launcher of WORLD_SIZE {world}: start it with --nproc-per-node {args.gpus}')
    if os.environ.get('TLLM_TEST_SHARED_GPU') == '1':
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if os.environ.get('TLLM_TEST_SHARED_GPU') == '1':
            # test rig (tests/test_bench_multirank.py): N ranks share GPU 0, gloo carries torch.distributed, the library's
            # peer-to-peer transport carries the model's collectives - exercises this file's N > 1 flow on a 1-GPU box
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        # TP communicator for the plugin library: rank 0's RCCL unique id -> everyone (replaces the MPI bootstrap), then
        # the one-shot peer-to-peer all-reduce if - and only if - it reproduces RCCL's sums on this node
        from tensorrt_llm import Mapping
        from tensorrt_llm.parallel import enable_p2p_allreduce, ensure_tp_communicator
        mapping = Mapping(world, rank)
        ensure_tp_communicator(mapping)
        import ctypes
        from tensorrt_llm.plugin import capi
        lib = capi.load_library()
        p2p_ok = enable_p2p_allreduce(mapping)
        lib.tllm_comm_p2p_state.restype = ctypes.c_int32
        fused_ok = p2p_ok and bool(lib.tllm_comm_p2p_state() & 4) and not os.environ.get('TLLM_NO_FUSED_ALLREDUCE')
        # Every transport that is available on this node is timed in this one run (the first contact with
        # xGMI must yield a scaling curve even if the hand-written transport fails):
        #   rccl       ncclAllReduce inside the step graph, three-stage seam (the reference's: allreducePlugin.cpp:80-96)
        #   p2p        the one-shot peer-to-peer kernel, three-stage seam
        #   p2p_fused  all-reduce + residual add + next RMSNorm (+ quantiser) in one launch per seam
        # `value` is the fastest leg whose outputs are finite; config.transports carries all of them.
        transports = []
        if os.environ.get('TLLM_TEST_SHARED_GPU') != '1':
            transports.append('rccl')
        if p2p_ok:
            transports.append('p2p')
        if fused_ok:
            transports.append('p2p_fused')
        allreduce_path = transports[-1]
        grp = (ctypes.c_int32 * world)(*mapping.tp_group)
        nr, me = ctypes.c_int32(0), ctypes.c_int32(-1)
        lib.tllm_comm_group_info.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        rccl_ranks = int(nr.value) if lib.tllm_comm_group_info(grp, world, ctypes.byref(nr), ctypes.byref(me)) == 0 else None
    else:
        allreduce_path = None
        rccl_ranks = None

    # Everything runs on a torch stream of its own, not on the legacy default stream: handed the NULL stream, the session
    # replays its graph on a private stream, and a timing event recorded on the NULL stream next to those replays (round 1's
    # device-time pair) drags legacy-stream ordering between the two - 9 % of GPU time (1.78 vs 1.62 ms per step, measured
    # both ways on one box).  On one real stream an event pair costs nothing (1.625 vs 1.619).
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    transport_results = {}
    if world > 1:
        lib.tllm_comm_p2p_enable.argtypes = [ctypes.c_int32]
        lib.tllm_comm_p2p_enable.restype = ctypes.c_int32
        lib.tllm_comm_p2p_enable_fused.argtypes = [ctypes.c_int32]
        lib.tllm_comm_p2p_enable_fused.restype = None

        def select(tr):
            rc = lib.tllm_comm_p2p_enable(0 if tr == 'rccl' else 1)
            lib.tllm_comm_p2p_enable_fused(1 if tr == 'p2p_fused' else 0)
            return rc == 0

        res = None
        for tr in transports:
            r, err = None, ''
            try:
                if not select(tr):
                    raise RuntimeError(capi.last_error())
                r = run_config(torch, dist, args, args.config, rank, world, dev)
                if not r['finite']:
                    r, err = None, 'non-finite logits'
            except Exception as e:  # a failed leg must not cost the curve: the verdict is made collective, the next leg runs
                r, err = None, repr(e)
            ok = torch.tensor([1 if r is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) != 1:
                transport_results[tr] = {'error': err or 'failed on another rank'}
                print(f'[bench rank {rank}] transport {tr} failed: {err}', file=sys.stderr, flush=True)
                continue
            transport_results[tr] = {'tokens_per_s': r['tokens_per_s'], 'ms_per_step': r['ms_per_step'],
                                     'comm_us_per_step': r['profile']['comm'][0] * 1e3 / 8,
                                     'comm_launches_per_step': r['profile']['comm'][1] / 8,
                                     'step_launch': 'hipGraph replay' if r.get('graph', True) else 'eager'}
            if res is None or r['tokens_per_s'] > res['tokens_per_s']:
                res, allreduce_path = r, tr
        if res is None:
            raise SystemExit(f'bench.py: no transport completed the run: {transport_results}')
        # the winner decides on every rank alike (tokens_per_s is the max-over-ranks wall time: identical everywhere)
        select(allreduce_path)
    else:
        res = run_config(torch, dist, args, args.config, rank, world, dev)
    fp16 = woq8 = None
    if args.config != 'fp16' and not args.no_fp16_ref:
        fp16 = run_config(torch, dist, args, 'fp16', rank, world, dev)
        if args.config == 'sq':  # BASELINE.json configs[2] next to configs[1] and [3]: weight-only int8 + int8 KV (side report)
            woq8 = run_config(torch, dist, args, 'woq8', rank, world, dev)
    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    prof = res['profile']
    prof_steps = 8
    # PMC counters cannot be read from inside the timed run: `traffic` is the committed rocprofv3 measurement of the same
    # kernel / shape (tools/refresh_pmc.sh -> profiles/*_pmc_summary.json), named in `traffic_source` - not this run's
    pmc_rows, traffic_source = [], None
    for rnd in ('r06', 'r05', 'r04', 'r03', 'r02', 'r01'):
        pmc = os.path.join(ROOT, 'profiles', f'{rnd}_pmc_summary.json')
        if os.path.exists(pmc):
            try:
                pmc_rows = json.load(open(pmc)).get(args.config, {}).get('kernels') or []
            except Exception:
                pmc_rows = []
            if pmc_rows:
                traffic_source = f'profiles/{rnd}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE passes, not this run)'
                break
    kernels = []
    for ln in res['launches']:
        us = ln['avg_us']
        gbs = ln['bytes'] / (us * 1e-6) / 1e9 if us > 0 else 0.0
        tr, tr_kernel = pmc_traffic(ln, pmc_rows)
        kernels.append({'kernel': ln['kernel'], 'what': ln['what'], 'launches_per_step': ln['calls'], 'avg_launch_us': us,
                        'us_per_step': us * ln['calls'], 'bytes_per_launch': ln['bytes'], 'achieved': gbs, 'frac': gbs / HBM_PEAK_GBS,
                        'traffic': tr, 'traffic_kernel': tr_kernel})
    dom = max(kernels, key=lambda k: k['us_per_step'])  # the dominant launch of THIS run: max(calls x average duration)
    cpu = None
    parity = cpu_model = cpu_info = None
    if world == 1 and args.parity and args.config == 'sq':
        # the accuracy half of the metric, on identical weights, at BASELINE.json configs[0]'s shape (bench_parity.py)
        try:
            import bench_parity
            parity, cpu_model, cpu_info = bench_parity.run(
                torch, dev, layers=args.layers, new_tokens=args.parity_new_tokens, n_prompts=args.parity_prompts,
                cpu_threads=args.cpu_threads or _default_cpu_threads(), cpu_leg=not args.no_cpu_baseline,
                log=lambda m: print(f'[bench parity] {m}', file=sys.stderr, flush=True))
        except Exception as e:  # the decode metric must not depend on the side report
            import traceback
            traceback.print_exc(file=sys.stderr)
            parity = {'error': repr(e)}
        # ... and on the TRAINED parent, where "ROUGE-L delta vs HF <= 1" is decidable (tests/golden/trained_llama): the product's
        # own convert -> build -> summarize command lines, six configurations, ~1 minute
        try:
            parity['trained_parent'] = bench_parity.trained_parent_report(
                log=lambda m: print(f'[bench parity] {m}', file=sys.stderr, flush=True))
        except Exception as e:
            parity['trained_parent'] = {'error': repr(e)}
    if world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(args.cpu_tokens, args.context, args.cpu_threads,
                               [int(x) for x in args.cpu_sweep.split(',') if x], model=cpu_model, layers=args.layers)
            if cpu_info:
                # BASELINE.json configs[0]: HF CPU, batch 1, prompt 128 - the reference's own latency definition
                # (run_hf.py: generate() wall time; tokens/s = new tokens / latency, T/benchmarks/gpt_benchmark.py:339)
                cpu['config0_hf_cpu_prompt128'] = {
                    'tokens_per_s': cpu_info['tokens_per_s'], 'latency_s': cpu_info['latency_s'], 'prompt_len': cpu_info['prompt_len'],
                    'new_tokens': cpu_info['new_tokens'], 'threads': cpu_info['threads'],
                    'definition': 'new tokens / generate() latency, prefill included (run_hf.py semantics)'}
        except Exception as e:
            cpu = dict(value=None, unit='tokens/s', cores=os.cpu_count(), kind='reference', sample=f'failed: {e!r}')
    del cpu_model
    names = {'sq': 'SmoothQuant per-channel int8 (act+weight) + int8 KV cache', 'woq8': 'weight-only int8 + int8 KV cache',
             'woq4': 'weight-only int4 + int8 KV cache', 'fp16': 'fp16 + fp16 KV cache'}
    line = {
        'metric': 'decode tokens/sec LLaMA-7B int8' if args.config != 'fp16' else 'decode tokens/sec LLaMA-7B fp16',
        'value': res['tokens_per_s'],
        'unit': 'tokens/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': res['ms_per_step'],
        'higher_is_better': True,
        'scaling': 'strong',
        'vs_baseline': None,
        'dtype': {'sq': 'int8', 'woq8': 'f16', 'woq4': 'f16', 'fp16': 'f16'}[args.config],
        'data': 'synthetic',
        'config': {'workload': f'LLaMA-7B ({args.layers} layers) {names[args.config]}, batch 1, context {args.context} '
                               f'(synthetic KV), greedy decode, TP={world}', 'global_batch': 1,
                   'seq_len': args.context, 'parallelism': f'tp{world}', 'allreduce': allreduce_path,
                   'transports': transport_results or None,
                   'rccl_communicator_ranks': rccl_ranks,
                   'comm_us_per_step': (prof['comm'][0] * 1e3 / prof_steps) if world > 1 else 0.0,
                   'comm_launches_per_step': (prof['comm'][1] / prof_steps) if world > 1 else 0,
                   'step_launch': 'hipGraph replay' if res.get('graph', True) else 'eager (graph capture failed)'},
        'roofline': {'bound': 'hbm', 'achieved': dom['achieved'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': dom['frac'],
                     'traffic': dom['traffic'], 'traffic_source': traffic_source if dom['traffic'] is not None else None,
                     # name as rocprofv3 prints it; the dominant launch = max(launches x average duration) over `kernels`
                     'kernel': '%s (%s; %.0f %% of the step\'s launches)'
                               % (dom['kernel'], dom['what'], 100.0 * dom['us_per_step'] / max(sum(k['us_per_step'] for k in kernels), 1e-9)),
                     'bytes_per_launch': dom['bytes_per_launch'], 'avg_launch_us': dom['avg_launch_us'],
                     'kernels': kernels},
        'cpu_baseline': cpu,
        'step': {'hbm_bytes': res['step_bytes'], 'hbm_frac_of_peak': res['step_bytes'] / (res['ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 'device_ms_per_step': res['dev_ms'] / args.steps, 'outputs_finite': res['finite'],
                 # the launches one layer is made of in this run, timed one by one (the same numbers as roofline.kernels)
                 'layer_kernel_us': {k: v for k, v in res['kernel_us'].items() if k != 'head'},
                 'decode_form': {'qkv_and_attention_in_one_launch': bool(res['decode_form'] & 1),
                                 'o_projection_in_that_launch': bool(res['decode_form'] & 2),
                                 'mlp_in_one_launch': bool(res['decode_form'] & 4)},
                 # the layer as the graph replay runs it: (device time per step - the head GEMV - the sampler, both from the eager
                 # profile below) / layers.  layer_kernel_us above times the launches ONE BY ONE
                 'layer_us_in_graph_replay': (res['dev_ms'] / args.steps - (prof['gemv_head'][0] + prof['other'][0]) / prof_steps) * 1e3 / args.layers,
                 'profile_ms_per_step': {k: v[0] / prof_steps for k, v in prof.items()},
                 'launches_per_step': {k: v[1] / prof_steps for k, v in prof.items()}},
    }
    if 'prefill_ms' in res:
        out_len = 128  # the reference's example run: 1024-token prompt, 128 new tokens
        line['prefill'] = {'context': args.context, 'ms': res['prefill_ms'],
                           'prompt_tokens_per_s': args.context / (res['prefill_ms'] * 1e-3),
                           # the reference's own benchmark definition folds the prefill in: B * outlen / latency
                           # (T/benchmarks/gpt_benchmark.py:339) - derived from the two measured phases
                           'reference_definition_tokens_per_s': out_len / ((res['prefill_ms'] + out_len * res['ms_per_step']) * 1e-3),
                           'reference_definition_output_len': out_len}
    if 'batch' in res:
        line['decode_over_batch'] = {'note': 'side report, same step with B sequences (graph replay); the metric above is batch 1',
                                     'context': args.context, **res['batch']}
    if args.prefill and args.config == 'sq' and world == 1:
        try:
            line['sq_gemm_mfma'] = sq_gemm_mfma_report(torch, dev)
            # the same four shapes at M = 4096 and 8192 (build.py's default max_batch_size 8 x 1024-token prompts): with several
            # rounds of workgroups the loop rate and the fixed cost of a launch (first-tile latency, exposed epilogue of the last
            # round) separate - at M = 1024 one round of 256 workgroups exposes all of it
            line['sq_gemm_mfma_large_m'] = {
                str(m): {k: {kk: v[kk] for kk in ('us', 'TOP/s', 'frac_of_5POPs', 'tactic', 'static_rule_us', 'shader_MHz_held',
                                                   'frac_of_peak_at_held_clock') if kk in v}
                         for k, v in sq_gemm_mfma_report(torch, dev, M=m).items()} for m in (2048, 4096, 8192)}
        except Exception as e:  # the decode metric must not depend on the side report
            line.setdefault('sq_gemm_mfma', {'error': repr(e)})
            line['sq_gemm_mfma_large_m'] = {'error': repr(e)}
    if world == 1 and args.config == 'sq' and getattr(args, 'tp_prediction', True):
        line['tp_rank_prediction'] = tp_rank_prediction(torch, args, dev)
    if parity is not None:
        line['parity'] = parity
    if fp16 is not None:
        line['fp16_tokens_per_s'] = fp16['tokens_per_s']
        if woq8 is not None:
            line['woq8_tokens_per_s'] = woq8['tokens_per_s']
        line['int8_over_fp16'] = res['tokens_per_s'] / fp16['tokens_per_s']
    print(json.dumps(line))
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
