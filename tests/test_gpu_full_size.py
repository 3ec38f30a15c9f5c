"""BASELINE.json's full size - LLaMA-7B, 32 layers, the benchmark's synthetic weights - through size-independent properties
(an oracle run of a 7B model does not finish in seconds):
  * replay invariance: the captured step graph reproduces eager launches token for token;
  * batch invariance: a prompt decoded alone and the same prompt twice in a batch of two give the same tokens.  For
    SmoothQuant (with the one-launch QKV projection + attention off) the two runs take DIFFERENT kernels for the single-token
    projections (batch 1: the K-split one-shot kernel for the down-projection, gemv_ksplit.hip; batch 2: the general kernel's
    two-row variant) whose int32 sums are exact, so equality here cross-checks them at the real shapes on all 32 layers;
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
    s._weights = w
    return s, cfg


@pytest.mark.parametrize('mode', ['sq', 'fp16', 'woq8', 'woq4'])
def test_full_size_replay_and_batch_invariance(mode):
    s, cfg = session(mode)
    w_ = s._weights
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
    # batch 1 and batch 2 take different kernels (batch 1: the one-launch QKV projection + attention and the K-split
    # down-projection; batch 2: the two launches and the general kernel's two-row variant): the integer GEMVs are exact, the fp32
    # order inside the attention differs - a near-tie of this random-weight model may flip
    assert np.mean(out2[0, S:] == out_graph[0, S:]) > 0.8
    if mode == 'sq':  # exact integer sums everywhere when both runs take the two launches
        s1 = NativeSession(dict(cfg, quant_mode=bench.QM[mode] | bench.INT8_KV, tp_size=1, tp_rank=0, fuse_qkv_attention=0))
        for k, v in w_.items():
            s1.set_tensor(k, v)
        s1.finalize()
        s1.setup(1, S, NEW)
        one = s1.generate(ids, lens, NEW)
        s1.setup(2, S, NEW)
        two = s1.generate(np.repeat(ids, 2, 0), np.repeat(lens, 2), NEW)
        np.testing.assert_array_equal(two[0], one[0])
        s1.close()
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


@pytest.mark.parametrize('fuse', [1, 0])
def test_in_launch_hand_offs_under_uneven_load(fuse):
    """The in-launch hand-offs - fuse = 1: the tagged granules of the one-launch QKV projection + attention (qkv_attn_fused.hip: q
    inside a head, partials + k, v to the head's merger); fuse = 0: the split merge of the attention launch (mmha_decode.hip step 6:
    write-through partials -> drain -> ticket -> agent-scope loads) - must hold when the chip is NOT idle: the same SmoothQuant
    generation (32 layers, 1100-token context, graph replay) with a second stream streaming 512 MB copies through HBM and L2 the
    whole time - arrival order, store latency and cache state all differ from the quiet run - must give the quiet run's tokens and
    logits bit for bit (a stale, torn or early-read granule / partial would change a logit)."""
    import threading
    cfg = dict(bench.LLAMA_7B)
    dev = torch.device('cuda', 0)
    w = bench.synth_weights(torch, cfg, 'sq', True, 1, 0, dev)
    s = NativeSession(dict(cfg, quant_mode=bench.QM['sq'] | bench.INT8_KV, tp_size=1, tp_rank=0, fuse_qkv_attention=-1 if fuse else 0))
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
