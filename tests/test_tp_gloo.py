"""Tensor-parallel path on CPU, world_size 2 over gloo: the host-side sharding the loaders do
(examples/llama_quant/weight.py: split / split_qkv / vocab padding — reference T/examples/llama/weight.py:86-140,
T/tensorrt_llm/_utils.py:194-195) plus the exchange steps the C++ session places (one sum all-reduce after the
row-parallel O projection and after the row-parallel down projection, rank 0 carrying the residual; all-gather of the
vocab-split logits — reference Q/llama_model.py:78-119, PY/layers/linear.py:118-134, :61-77) must reproduce the
un-sharded model.  Each rank runs the numpy oracle on ITS shard only; the collectives are real (gloo)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant')
GOLD = os.path.join(ROOT, 'tests', 'golden')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_weights(t, tp, rank, n_layers, vocab):
    """golden (reference module naming, un-sharded) -> this rank's oracle weight dict, via the loaders' helpers."""
    sys.path.insert(0, EX)
    import weight as W  # examples/llama_quant/weight.py
    f32 = lambda a: np.asarray(a, np.float32)
    head = t['lm_head.weight'][:vocab]
    vpad = -head.shape[0] % tp
    if vpad:
        head = np.pad(head, ((0, vpad), (0, 0)))
    ow = {'vocab_embedding.weight': f32(t['vocab_embedding.weight']), 'ln_f.weight': f32(t['ln_f.weight']),
          'lm_head.weight': f32(W.split(head, tp, rank)), 'layers': []}
    for i in range(n_layers):
        p = f'layers.{i}.'
        ow['layers'].append({
            'input_layernorm.weight': f32(t[p + 'input_layernorm.weight']),
            'post_layernorm.weight': f32(t[p + 'post_layernorm.weight']),
            'attention.qkv.weight': f32(W.split_qkv(t[p + 'attention.qkv.weight'], tp, rank)),
            'attention.dense.weight': f32(W.split(t[p + 'attention.dense.weight'], tp, rank, dim=1)),
            'mlp.fc.weight': f32(W.split(t[p + 'mlp.fc.weight'], tp, rank, dim=0)),
            'mlp.gate.weight': f32(W.split(t[p + 'mlp.gate.weight'], tp, rank, dim=0)),
            'mlp.proj.weight': f32(W.split(t[p + 'mlp.proj.weight'], tp, rank, dim=1)),
        })
    return ow


def _tp_forward(rank, world, port, vocab, out_q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import llama_oracle as O
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        t = dict(np.load(os.path.join(GOLD, 'hf_tiny_llama.npz')))
        L, H, D = 2, 2, 64
        Hr, Dh = H // world, D // H
        ids, lens = t['ids'], t['input_lengths']
        B, S = ids.shape
        smax = S + 4
        w = _shard_weights(t, world, rank, L, vocab)

        def allreduce16(partial):
            # the session all-reduces fp16 partials (ncclAllReduce sum, fp16); the sum of two fp16 values rounds once
            x = torch.from_numpy(np.ascontiguousarray(O.f16(partial), dtype=np.float32))
            dist.all_reduce(x)
            return O.f16(x.numpy())

        def head(x_last):
            part = (x_last @ w['lm_head.weight'].T).astype(np.float32)  # [B, Vr]
            buf = [torch.zeros(part.shape, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(buf, torch.from_numpy(np.ascontiguousarray(part)))
            return np.concatenate([b.numpy() for b in buf], axis=1)[:, :vocab]  # drop the vocab padding

        caches = [np.zeros((B, 2, Hr, smax, Dh), np.float16) for _ in range(L)]
        # ---- context phase
        x = O.f16(w['vocab_embedding.weight'][ids])
        for li, lw in enumerate(w['layers']):
            h = O.rmsnorm(x, lw['input_layernorm.weight'])
            qkv = O.gemm_fp16(h.reshape(B * S, D), lw['attention.qkv.weight']).reshape(B, S, 3 * Hr * Dh)
            ctx, _ = O.context_attention(qkv, caches[li], lens, Hr, Dh, Dh, True, 1.0, None)
            part = O.gemm_fp16(ctx.reshape(B * S, Hr * Dh), lw['attention.dense.weight']).reshape(B, S, D)
            x = allreduce16(part + (x if rank == 0 else 0.0))  # rank 0 carries the residual into the sum
            h2 = O.rmsnorm(x, lw['post_layernorm.weight'])
            g = O.gemm_fp16(h2.reshape(B * S, D), lw['mlp.fc.weight'])
            u = O.gemm_fp16(h2.reshape(B * S, D), lw['mlp.gate.weight'])
            part = O.gemm_fp16(O.swiglu(g, u), lw['mlp.proj.weight']).reshape(B, S, D)
            x = allreduce16(part + (x if rank == 0 else 0.0))
        xn = O.rmsnorm(x, w['ln_f.weight'])
        last = np.stack([xn[b, int(lens[b]) - 1] for b in range(B)])
        logits_ctx = head(last)
        nxt = logits_ctx.argmax(-1).astype(np.int32)
        # ---- one generation step
        masked = np.zeros((B, smax), np.int32)
        for b in range(B):
            masked[b, lens[b]:S] = 1
        x = O.f16(w['vocab_embedding.weight'][nxt])
        for li, lw in enumerate(w['layers']):
            h = O.rmsnorm(x, lw['input_layernorm.weight'])
            qkv = O.gemm_fp16(h, lw['attention.qkv.weight'])
            ctx = O.mmha_decode(qkv, caches[li], [S] * B, lens, S, S, Hr, Dh, Dh, True, 1.0, masked)
            x = allreduce16(O.gemm_fp16(ctx, lw['attention.dense.weight']) + (x if rank == 0 else 0.0))
            h2 = O.rmsnorm(x, lw['post_layernorm.weight'])
            g = O.gemm_fp16(h2, lw['mlp.fc.weight'])
            u = O.gemm_fp16(h2, lw['mlp.gate.weight'])
            x = allreduce16(O.gemm_fp16(O.swiglu(g, u), lw['mlp.proj.weight']) + (x if rank == 0 else 0.0))
        logits_dec = head(O.rmsnorm(x, w['ln_f.weight']))
        # every rank must hold identical results after the collectives
        chk = torch.from_numpy(np.concatenate([logits_ctx.ravel(), logits_dec.ravel()]).astype(np.float64))
        mx = chk.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        same = bool((mx == chk).all())
        if rank == 0:
            out_q.put((logits_ctx, nxt, logits_dec, same, [c.copy() for c in caches]))
        else:
            out_q.put(('rank1', same, [c.copy() for c in caches]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('vocab', [128, 127])  # 127: the vocabulary is padded to a multiple of tp
def test_tp2_matches_unsharded(vocab):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from oracle import llama_oracle as O
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tp_forward, args=(r, 2, port, vocab, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = next(g for g in got if not isinstance(g[0], str))
    r1 = next(g for g in got if isinstance(g[0], str))
    logits_ctx, nxt, logits_dec, same0, caches0 = r0
    assert same0 and r1[1], 'ranks disagree after the collectives'

    # the un-sharded oracle (which tests/test_oracle_golden.py pins to the HF golden values)
    t = dict(np.load(os.path.join(GOLD, 'hf_tiny_llama.npz')))
    ids, lens = t['ids'], t['input_lengths']
    B, S = ids.shape
    H, Dh, smax = 2, 32, S + 4
    ow = _shard_weights(t, 1, 0, 2, vocab)
    caches = [np.zeros((B, 2, H, smax, Dh), np.float16) for _ in range(2)]
    ref_ctx = O.llama_logits_context(ids, ow, caches, lens, H)
    # fp16 partial sums are rounded per rank before the all-reduce: not bit-identical, well inside the fp16 bound
    np.testing.assert_allclose(logits_ctx, ref_ctx, atol=2e-2)
    np.testing.assert_allclose(logits_ctx, t['logits_ctx'][:, :vocab], atol=1e-1)  # HF golden, reference bound
    np.testing.assert_array_equal(nxt, ref_ctx.argmax(-1))
    masked = np.zeros((B, smax), np.int32)
    for b in range(B):
        masked[b, lens[b]:S] = 1
    ref_dec = O.llama_logits_decode(nxt, ow, caches, [S, S], lens, S, S, H, masked)
    np.testing.assert_allclose(logits_dec, ref_dec, atol=2e-2)
    # KV cache is sharded by head: rank r holds heads [r*Hr, (r+1)*Hr) of the un-sharded cache (layer 0 is exact:
    # nothing upstream of it has been all-reduced)
    np.testing.assert_array_equal(caches0[0][:, :, 0], caches[0][:, :, 0])
    np.testing.assert_array_equal(r1[2][0][:, :, 0], caches[0][:, :, 1])
