"""BASELINE.json's full size - LLaMA-7B, 32 layers, the benchmark's synthetic weights - through size-independent properties
(an oracle run of a 7B model does not finish in seconds):
  * replay invariance: the captured step graph reproduces eager launches token for token;
  * batch invariance: a prompt decoded alone and the same prompt twice in a batch of two give the same tokens.  For
    SmoothQuant the two runs take DIFFERENT kernels for the single-token projections (batch 1: the K-split one-shot kernel for
    the down-projection, gemv_ksplit.hip; batch 2: the general kernel's two-row variant) whose int32 sums are exact, so
    equality here cross-checks them at the real shapes on all 32 layers;
  * padding invariance: the logits of a prompt do not depend on how far its buffer is padded (max_input_len)."""
import numpy as np
import pytest
import torch

import bench
from tensorrt_llm.runtime.native import NativeSession

pytestmark = pytest.mark.gpu


def session(mode, **extra):
    cfg = dict(bench.LLAMA_7B)
    int8_kv = mode != 'fp16'
    dev = torch.device('cuda', 0)
    s = NativeSession(dict(cfg, quant_mode=bench.QM[mode] | (bench.INT8_KV if int8_kv else 0), tp_size=1, tp_rank=0, **extra))
    w = bench.synth_weights(torch, cfg, mode, int8_kv, 1, 0, dev)
    for k, v in w.items():
        s.set_tensor(k, v)
    s.finalize()
    return s, cfg


@pytest.mark.parametrize('mode', ['sq', 'fp16', 'woq8', 'woq4'])
def test_full_size_replay_and_batch_invariance(mode):
    s, cfg = session(mode)
    S, NEW = 96, 20
    r = np.random.default_rng(17)
    ids = r.integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    lens = np.array([S], np.int32)
    # generate(): first step eager, the rest replayed from the graph
    s.setup(1, S, NEW)
    out_graph = s.generate(ids, lens, NEW)
    logits_graph = s.logits()
    # all steps eager
    s.setup(1, S, NEW)
    s.context(ids, lens)
    s.step(NEW - 1, use_graph=False)
    out_eager = s.output_ids()
    np.testing.assert_array_equal(out_graph, out_eager)
    np.testing.assert_array_equal(s.logits(), logits_graph)  # the whole distribution of the last step, not only its arg-max
    if mode in ('sq', 'fp16'):  # (the int4 synthetic model repeats one token: there the logits carry the comparison)
        assert len(set(out_graph[0, S:].tolist())) > 1, 'degenerate generation: the token comparison would prove nothing'
    # the same prompt twice in a batch of two
    s.setup(2, S, NEW)
    out2 = s.generate(np.repeat(ids, 2, 0), np.repeat(lens, 2), NEW)
    np.testing.assert_array_equal(out2[0], out2[1])
    if mode == 'sq':  # exact integer sums: also identical to the batch-1 run (fp16 sums may differ in the last bit)
        np.testing.assert_array_equal(out2[0], out_graph[0])
    else:
        assert np.mean(out2[0, S:] == out_graph[0, S:]) > 0.8
    s.close()


def test_full_size_context_logits_do_not_depend_on_the_padding():
    s, cfg = session('fp16')
    S = 64
    r = np.random.default_rng(5)
    ids = r.integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    s.setup(1, S, 4)
    s.context(ids, np.array([S], np.int32))
    full = s.logits()
    s.setup(1, S + 32, 4)  # the same prompt in a longer, right-padded buffer: padding rows are masked
    padded = np.full((1, S + 32), 2, np.int32)
    padded[0, :S] = ids[0]
    s.context(padded, np.array([S], np.int32))
    again = s.logits()
    scale = max(np.abs(full).max(), 1.0)
    np.testing.assert_allclose(again, full, atol=2e-2 * scale)  # a different tile split of the GEMMs: fp16 summation order only
    assert int(again.argmax()) == int(full.argmax())
    s.close()


@pytest.mark.parametrize('beam', [1, 3])
def test_full_size_paged_cache_equals_linear(beam):
    """Paged KV cache (64-token blocks) at the full model size: same tokens as the linear cache, greedy and beam search, with
    a prompt that ends in the middle of a block."""
    S, NEW = 150, 12
    r = np.random.default_rng(29)
    ids = r.integers(3, 32000, (1, S)).astype(np.int32)
    lens = np.array([S], np.int32)
    outs = []
    for paged in (0, 1):
        s, _ = session('sq', paged_kv_cache=paged, tokens_per_block=64)
        s.setup(1, S, NEW, beam_width=beam)
        outs.append(s.generate(ids, lens, NEW))
        s.close()
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize('mode,layers', [('sq', 4), ('sq', 32), ('fp16', 4), ('woq8', 4)])
def test_in_launch_attention_merge_equals_the_prologue_merge(mode, layers, monkeypatch):
    """The split-KV merge inside the attention launch (r04, mmha_decode.hip step 6: write-through partials, one ticket per
    workgroup, the last arriver of a head merges with agent-scope loads) against the r01 - r03 path (TLLM_NO_ATTN_TAIL_MERGE=1:
    every O-projection workgroup merges all partials in its prologue).  Same slot order, same fp32 arithmetic, and for SmoothQuant
    exact integer GEMVs behind it: tokens and logits must be IDENTICAL, eager and replayed from the graph, over contexts that use
    1 ... 7 splits, on 4 and on all 32 layers (the partial buffers are rewritten by every layer of every step: a stale or torn partial
    would show up as a wrong logit).  fp16 / weight-only: the O-projection behind a plain fp16 vector is the K-split kernel instead
    of the general one - same tokens, logits to fp32 summation order."""
    cfg = dict(bench.LLAMA_7B, num_layers=layers)
    int8_kv = mode != 'fp16'
    dev = torch.device('cuda', 0)
    w = bench.synth_weights(torch, cfg, mode, int8_kv, 1, 0, dev)
    results = {}
    for tail in (False, True):
        if tail:
            monkeypatch.delenv('TLLM_NO_ATTN_TAIL_MERGE', raising=False)
        else:
            monkeypatch.setenv('TLLM_NO_ATTN_TAIL_MERGE', '1')
        s = NativeSession(dict(cfg, quant_mode=bench.QM[mode] | (bench.INT8_KV if int8_kv else 0), tp_size=1, tp_rank=0))
        for k, v in w.items():
            s.set_tensor(k, v)
        s.finalize()
        outs = []
        for S, NEW in ((40, 12), (700, 24), (1100, 40)):
            ids = np.random.default_rng(S).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
            s.setup(1, S, NEW)
            toks = s.generate(ids, np.array([S], np.int32), NEW)  # first step eager, the rest from the graph
            outs.append((toks.copy(), s.logits().copy()))
        results[tail] = outs
        s.close()
    for (t0, l0), (t1, l1) in zip(results[False], results[True]):
        if mode == 'sq':
            np.testing.assert_array_equal(t0, t1)
            np.testing.assert_array_equal(l0, l1)
        else:
            assert np.mean(t0 == t1) > 0.9  # random weights: a near-tie may flip on the last bit
            if np.array_equal(t0, t1):
                np.testing.assert_allclose(l0, l1, atol=1e-2)


@pytest.mark.parametrize('mode', ['fp16', 'woq8', 'woq4'])
def test_prefill_epilogue_fusions_are_bit_identical(mode, monkeypatch):
    """r04 prefill fusions of the fp16 / weight-only paths against the separate passes they replace: SwiGLU folded into the second
    MLP projection's epilogue (GemmParams::silu_gate; TLLM_NO_SWIGLU_FUSE=1 runs swiglu_kernel) and the residual add folded into
    the weight-only O / down GEMMs (TLLM_NO_WOQ_RESIDUAL_FUSE=1 runs add_kernel).  Same rounding points (fp16(gemm), fp16(silu),
    fp16(product) / fp16(sum)): the context logits and the first greedy tokens must be IDENTICAL, at a prefill of one full
    workgroup round (1024 tokens) and a ragged one (333)."""
    cfg = dict(bench.LLAMA_7B, num_layers=3)
    int8_kv = mode != 'fp16'
    dev = torch.device('cuda', 0)
    w = bench.synth_weights(torch, cfg, mode, int8_kv, 1, 0, dev)
    results = {}
    for fused in (False, True):
        for k in ('TLLM_NO_SWIGLU_FUSE', 'TLLM_NO_WOQ_RESIDUAL_FUSE'):
            if fused:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, '1')
        s = NativeSession(dict(cfg, quant_mode=bench.QM[mode] | (bench.INT8_KV if int8_kv else 0), tp_size=1, tp_rank=0))
        for k, v in w.items():
            s.set_tensor(k, v)
        s.finalize()
        outs = []
        for S in (1024, 333):
            ids = np.random.default_rng(S).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
            s.setup(1, S, 4)
            toks = s.generate(ids, np.array([S], np.int32), 4)
            outs.append((toks.copy(), s.logits().copy()))
        results[fused] = outs
        s.close()
    for (t0, l0), (t1, l1) in zip(results[False], results[True]):
        np.testing.assert_array_equal(t0, t1)
        np.testing.assert_array_equal(l0, l1)


@pytest.mark.parametrize('B', [5, 8])
def test_several_sequences_on_the_matrix_pipe_equal_the_skinny_kernel(B, lib):
    """Decode with 5 - 8 sequences runs the SmoothQuant layer GEMMs on the matrix pipe (kernels/gemv_mfma_sq.hip, the default from 5
    rows on); tllm_gemv_set_mfma_rows(0) keeps the skinny vector-ALU kernel.  The two are bit-identical stage by stage, so a whole
    generation - prefill, eager first step, graph-replayed steps, ragged prompt lengths - must give IDENTICAL tokens and logits at the
    7B layer dimensions."""
    import ctypes
    lib.tllm_gemv_set_mfma_rows.argtypes = [ctypes.c_int32]
    lib.tllm_gemv_set_mfma_rows.restype = None
    cfg = dict(bench.LLAMA_7B, num_layers=3)
    dev = torch.device('cuda', 0)
    w = bench.synth_weights(torch, cfg, 'sq', True, 1, 0, dev)
    rng = np.random.default_rng(B)
    S, NEW = 96, 12
    lens = rng.integers(S // 2, S + 1, B).astype(np.int32)
    ids = rng.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    outs = []
    try:
        for rows_from in (0, -1):
            lib.tllm_gemv_set_mfma_rows(rows_from)
            s = NativeSession(dict(cfg, quant_mode=bench.QM['sq'] | bench.INT8_KV, tp_size=1, tp_rank=0))
            for k, v in w.items():
                s.set_tensor(k, v)
            s.finalize()
            s.setup(B, S, NEW)
            toks = s.generate(ids, lens, NEW)
            outs.append((toks.copy(), s.logits().copy()))
            s.close()
    finally:
        lib.tllm_gemv_set_mfma_rows(-1)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_in_launch_attention_merge_beyond_eight_partials():
    """More than 8 split partials (a cache of more than 2048 slots at head size 128): the in-launch merge takes up to 16, the
    prologue form stops at 8 and hands over to the finest split + combine launch - a different split, so fp32 summation order
    differs: same greedy tokens (or a flipped near-tie), logits close."""
    import os
    cfg = dict(bench.LLAMA_7B, num_layers=4, max_position_embeddings=4096)
    dev = torch.device('cuda', 0)
    w = bench.synth_weights(torch, cfg, 'sq', True, 1, 0, dev)
    S, NEW = 2300, 16
    ids = np.random.default_rng(3).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    res = []
    for tail in (True, False):
        if tail:
            os.environ.pop('TLLM_NO_ATTN_TAIL_MERGE', None)
        else:
            os.environ['TLLM_NO_ATTN_TAIL_MERGE'] = '1'
        try:
            s = NativeSession(dict(cfg, quant_mode=bench.QM['sq'] | bench.INT8_KV, tp_size=1, tp_rank=0))
            for k, v in w.items():
                s.set_tensor(k, v)
            s.finalize()
            s.setup(1, S, NEW)
            toks = s.generate(ids, np.array([S], np.int32), NEW)
            res.append((toks.copy(), s.logits().copy()))
            s.close()
        finally:
            os.environ.pop('TLLM_NO_ATTN_TAIL_MERGE', None)
    (t0, l0), (t1, l1) = res
    assert np.mean(t0[0, S:] == t1[0, S:]) >= 0.75
    if np.array_equal(t0, t1):
        np.testing.assert_allclose(l0, l1, atol=0.05 * max(1.0, float(np.abs(l1).max())))


def test_in_launch_attention_merge_under_uneven_load():
    """The hand-off of the in-launch merge (write-through partials -> drain -> ticket -> agent-scope loads) must hold when the chip
    is NOT idle: the same SmoothQuant generation (32 layers, 1100-token context, graph replay) with a second stream streaming
    512 MB copies through HBM and L2 the whole time - arrival order, store latency and cache state all differ from the quiet run -
    must give the quiet run's tokens and logits bit for bit (a stale, torn or early-read partial would change a logit)."""
    import threading
    cfg = dict(bench.LLAMA_7B)
    dev = torch.device('cuda', 0)
    w = bench.synth_weights(torch, cfg, 'sq', True, 1, 0, dev)
    s = NativeSession(dict(cfg, quant_mode=bench.QM['sq'] | bench.INT8_KV, tp_size=1, tp_rank=0))
    for k, v in w.items():
        s.set_tensor(k, v)
    s.finalize()
    S, NEW = 1100, 48
    ids = np.random.default_rng(41).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    lens = np.array([S], np.int32)

    def run():
        s.setup(1, S, NEW)
        toks = s.generate(ids, lens, NEW)
        return toks.copy(), s.logits().copy()

    quiet = run()
    stop = threading.Event()
    side = torch.cuda.Stream(device=dev)
    a = torch.empty(128 << 20, dtype=torch.float32, device=dev)  # 512 MB
    b = torch.empty_like(a)

    def hammer():
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(8):
                    b.copy_(a, non_blocking=True)
                    a.add_(1.0)
                side.synchronize()

    th = threading.Thread(target=hammer, daemon=True)
    th.start()
    try:
        loaded = [run() for _ in range(3)]
    finally:
        stop.set()
        th.join(timeout=30)
    torch.cuda.synchronize()
    for toks, logits in loaded:
        np.testing.assert_array_equal(toks, quiet[0])
        np.testing.assert_array_equal(logits, quiet[1])
    s.close()


@pytest.mark.parametrize('B', [8, 12])
@pytest.mark.parametrize('mode', ['fp16', 'woq8'])
def test_batch_of_eight_at_7b_dimensions(mode, B):
    """build.py's default --max_batch_size 8 at the 7B layer dimensions with fp16 activations: the down-projection's 8 activation
    rows (K = 11008 halfs: 176 KB) do not fit a CU's LDS at once - the GEMV takes them in slabs of 4 (r04; refused before).  Every
    row of the batch is the same prompt: all eight sequences must generate what the batch-1 run generates."""
    cfg = dict(bench.LLAMA_7B, num_layers=2)
    int8_kv = mode != 'fp16'
    dev = torch.device('cuda', 0)
    s = NativeSession(dict(cfg, quant_mode=bench.QM[mode] | (bench.INT8_KV if int8_kv else 0), tp_size=1, tp_rank=0))
    for k, v in bench.synth_weights(torch, cfg, mode, int8_kv, 1, 0, dev).items():
        s.set_tensor(k, v)
    s.finalize()
    S, NEW = 48, 10
    ids = np.random.default_rng(9).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    s.setup(1, S, NEW)
    one = s.generate(ids, np.array([S], np.int32), NEW)
    s.setup(B, S, NEW)  # 12: the session's slabs of 8 rows (8 + 4), each split again by the GEMV where its rows exceed the LDS
    eight = s.generate(np.repeat(ids, B, 0), np.full(B, S, np.int32), NEW)
    s.close()
    for b in range(B):
        np.testing.assert_array_equal(eight[b], eight[0])
    assert np.mean(eight[0, S:] == one[0, S:]) > 0.7  # batch 1 and batch 8 take different kernels: fp32 summation order


@pytest.mark.parametrize('mode', ['sq', 'fp16'])
def test_maximum_prefill_batch8_x_2048(mode):
    """build.py's defaults at their limit: 8 prompts of 2048 tokens = 16384 rows through the prefill GEMMs (64 row tiles of 256; 32-bit
    LDS-DMA offsets checked against M * K), the context attention at its longest sequence on 8 x 32 heads, the KV write of
    8 x 2048 slots.  Every row of the batch is the same prompt: identical logits in all eight, equal (SmoothQuant: bit for bit, its
    GEMMs are exact) to the prompt run alone."""
    cfg = dict(bench.LLAMA_7B, num_layers=2)
    int8_kv = mode != 'fp16'
    dev = torch.device('cuda', 0)
    s = NativeSession(dict(cfg, quant_mode=bench.QM[mode] | (bench.INT8_KV if int8_kv else 0), tp_size=1, tp_rank=0))
    for k, v in bench.synth_weights(torch, cfg, mode, int8_kv, 1, 0, dev).items():
        s.set_tensor(k, v)
    s.finalize()
    S = 2044
    ids = np.random.default_rng(2).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    s.setup(1, S, 4)
    s.context(ids, np.array([S], np.int32))
    one = s.logits().copy()
    s.setup(8, S, 4)
    s.context(np.repeat(ids, 8, 0), np.full(8, S, np.int32))
    eight = s.logits().copy()
    s.step(2, use_graph=False)
    assert np.isfinite(s.logits()).all()
    s.close()
    assert np.isfinite(eight).all()
    for b in range(1, 8):
        np.testing.assert_array_equal(eight[b], eight[0])
    if mode == 'sq':
        np.testing.assert_array_equal(eight[0], one[0])
    else:
        np.testing.assert_allclose(eight[0], one[0], atol=2e-2 * max(1.0, float(np.abs(one).max())))
