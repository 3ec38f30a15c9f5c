"""GPU parity of the paged KV cache (K/kvCacheUtils.h:34-112 KVBlockArray,
P/gptAttentionPlugin/gptAttentionPlugin.cpp:313-325, PY/runtime/kv_cache_manager.py).

Paging only changes WHERE a K/V row lives, so the bar is bit-exactness against the linear cache: the same prompts through
the same kernels must give identical attention outputs, logits and tokens, and the pool - read back through the block
table - must hold exactly the linear cache's rows.  (The linear path itself is checked against the oracle in
test_gpu_plugins.py / test_gpu_session.py.)"""
import numpy as np
import pytest
import torch

from helpers import HostTensor, h, make_plugin, run_plugin
from helpers import i8, i32, f32
from tensorrt_llm.plugin import capi
from tensorrt_llm.runtime.kv_cache_manager import GenerationSequence, KVCacheManager
from tensorrt_llm.runtime.native import NativeSession
from test_gpu_session import synth_model
from oracle import quant_oracle as QO

pytestmark = pytest.mark.gpu


def attention_plugin(H, Dh, int8_kv, paged):
    return make_plugin('GPTAttention', [
        ('num_heads', i32(H)), ('head_size', i32(Dh)), ('unidirectional', i32(1)), ('q_scaling', f32(1.0)),
        ('rotary_embedding_dim', i32(Dh)), ('neox_rotary_style', i8(1)),
        ('context_fmha_type', i8(0)), ('multi_block_mode', i8(0)), ('multi_query_mode', i8(0)),
        ('int8_kv_cache', i32(int8_kv)), ('fp8_kv_cache', i32(0)), ('remove_input_padding', i8(0)),
        ('mask_type', i32([1])), ('paged_kv_cache', i32(paged)), ('type_id', i32([capi.HALF])), ('in_flight_batching', i32(0)),
    ])


def enqueue(p, qkv, cache, seq_len, past_len, is_context, masked, in_len, max_in, smax, scales, pointers=None, ci=None):
    B = qkv.shape[0]
    out = torch.empty(qkv.shape[:-1] + (qkv.shape[-1] // 3, ), dtype=torch.float16, device='cuda')
    dummy = torch.zeros(max(max_in, B * smax), dtype=torch.int32, device='cuda')
    ins = [qkv, cache, torch.tensor(seq_len, dtype=torch.int32, device='cuda'), HostTensor([past_len, 1 if is_context else 0]),
           torch.tensor(masked, dtype=torch.int32, device='cuda'), torch.tensor(in_len, dtype=torch.int32, device='cuda'),
           dummy[:max_in], dummy[:B * smax].view(B, 1, smax) if ci is None else ci]
    if scales is not None:
        ins += [torch.tensor([scales[0]], dtype=torch.float32, device='cuda'),
                torch.tensor([scales[1]], dtype=torch.float32, device='cuda')]
    if pointers is not None:
        ins += [pointers]
    run_plugin(p, ins, [out, cache])
    return out


def gather_pool(pool, table, H, T, Dh, smax):
    """pool -> the linear view [B, 2, H, smax, Dh] the block table (int64 [B, 1, 2, M]) describes."""
    flat = pool.reshape(-1).cpu().numpy()
    esz = flat.itemsize
    base = pool.data_ptr()
    B = table.shape[0]
    out = np.zeros((B, 2, H, smax, Dh), flat.dtype)
    for b in range(B):
        for kv in range(2):
            for t in range(smax):
                ptr = int(table[b, 0, kv, t // T])
                if ptr == 0:
                    continue
                blk = np.asarray(flat[(ptr - base) // esz:(ptr - base) // esz + H * T * Dh]).reshape(H, T, Dh)
                out[b, kv, :, t] = blk[:, t % T]
    return out


@pytest.mark.parametrize('int8_kv', [0, 1])
@pytest.mark.parametrize('H,Dh,T,S', [(4, 128, 64, 150), (2, 64, 16, 40), (4, 32, 128, 129)])
def test_paged_plugin_equals_linear(int8_kv, H, Dh, T, S):
    """Context phase then three generation steps through the plugin, linear cache vs pool + block pointers handed out by
    KVCacheManager (blocks of the two sequences interleave in the pool)."""
    r = np.random.default_rng(S)
    B, NEW = 2, 3
    smax = S + NEW
    in_len = [S, S // 2]
    masked = np.zeros((B, smax), np.int32)
    masked[1, in_len[1]:S] = 1
    scales = (20.0, 0.05) if int8_kv else None
    kv_dtype = torch.int8 if int8_kv else torch.float16
    M = -(-smax // T)
    blocks = B * M + 3
    pool = torch.zeros(blocks, 2, H, T, Dh, dtype=kv_dtype, device='cuda')
    lin = torch.zeros(B, 2, H, smax, Dh, dtype=kv_dtype, device='cuda')
    mgr = KVCacheManager([pool], blocks, T, M, beam_width=1)
    # interleave the allocations so that a sequence's blocks are scattered
    seqs = [GenerationSequence(b, b) for b in range(B)]
    for sq in seqs:
        mgr.add_sequence(sq, 0)
    while mgr.blocks_manager.get_number_blocks(seqs[0]) < M:
        for sq in seqs:
            mgr.blocks_manager.allocate(sq)
    table = mgr.blocks_manager.get_pointer_array(0)
    pointers = mgr.get_pointer_arrays()[0]
    assert pointers.shape == (B, 1, 2, 2 * M) and pointers.dtype == torch.int32
    p_lin, p_pg = attention_plugin(H, Dh, int8_kv, 0), attention_plugin(H, Dh, int8_kv, 1)
    qkv = h(r.standard_normal((B, S, 3 * H * Dh)))
    o1 = enqueue(p_lin, qkv.clone(), lin, [S, S], 0, True, masked, in_len, S, smax, scales)
    o2 = enqueue(p_pg, qkv.clone(), pool, [S, S], 0, True, masked, in_len, S, smax, scales, pointers)
    assert torch.equal(o1, o2)
    for step in range(NEW):
        q1 = h(r.standard_normal((B, 1, 3 * H * Dh)))
        L = S + step
        o1 = enqueue(p_lin, q1.clone(), lin, [L, L], L, False, masked, in_len, S, smax, scales)
        o2 = enqueue(p_pg, q1.clone(), pool, [L, L], L, False, masked, in_len, S, smax, scales, pointers)
        assert torch.equal(o1, o2), f'generation step {step}'
    np.testing.assert_array_equal(gather_pool(pool, table, H, T, Dh, smax), lin.cpu().numpy())


@pytest.mark.parametrize('int8_kv', [0, 1])
@pytest.mark.parametrize('H,Dh,T,S,NEW', [(4, 128, 16, 40, 41), (2, 64, 8, 21, 20)])
def test_paged_plugin_with_on_demand_block_allocation(int8_kv, H, Dh, T, S, NEW):
    """The flow INTEGRATION.md documents and the reference's GenerationSession runs (PY/runtime/generation.py:842-848,
    :978-983): KVCacheManager.add_sequence hands out only the blocks of the (padded) prompt plus one token, step() grows
    the table when a sequence crosses a block boundary - so for most of the run the table holds ZERO entries for the logical
    blocks beyond the current length, while the cache capacity the plugin sees (cache_indirection.shape[2]) covers them.
    The kernels must never form an address from such an entry: the split-KV generation kernel points the (masked) loads of
    those rows at the pool, the context KV write skips them.  Unequal prompt lengths; bit-identical to the linear cache."""
    r = np.random.default_rng(S + NEW)
    B = 2
    smax = S + NEW
    in_len = [S, S // 3]
    masked = np.zeros((B, smax), np.int32)
    masked[1, in_len[1]:S] = 1
    scales = (20.0, 0.05) if int8_kv else None
    kv_dtype = torch.int8 if int8_kv else torch.float16
    M = -(-smax // T)
    blocks = B * M + 2
    pool = torch.zeros(blocks, 2, H, T, Dh, dtype=kv_dtype, device='cuda')
    lin = torch.zeros(B, 2, H, smax, Dh, dtype=kv_dtype, device='cuda')
    mgr = KVCacheManager([pool], blocks, T, M, beam_width=1)
    for b in range(B):
        mgr.add_sequence(GenerationSequence(b, b), S)  # the padded prompt length for every sequence, as the reference does
    first = mgr.blocks_manager.get_number_blocks(mgr.sequences[0])
    assert first == -(-(S + 1) // T) and first < M, 'the scenario must leave logical blocks unallocated'
    pointers = mgr.get_pointer_arrays()[0]
    assert (mgr.blocks_manager.pointer_array[:, 0, :, first:] == 0).all()
    p_lin, p_pg = attention_plugin(H, Dh, int8_kv, 0), attention_plugin(H, Dh, int8_kv, 1)
    qkv = h(r.standard_normal((B, S, 3 * H * Dh)))
    o1 = enqueue(p_lin, qkv.clone(), lin, [S, S], 0, True, masked, in_len, S, smax, scales)
    o2 = enqueue(p_pg, qkv.clone(), pool, [S, S], 0, True, masked, in_len, S, smax, scales, pointers)
    assert torch.equal(o1, o2)
    grew = 0
    for step in range(NEW):
        L = S + step
        if step > 0:
            before = mgr.blocks_manager.get_number_blocks(mgr.sequences[0])
            mgr.step([False] * B)  # generation.py:978-983: after every step but the last
            grew += mgr.blocks_manager.get_number_blocks(mgr.sequences[0]) - before
            pointers = mgr.get_pointer_arrays()[0]
        assert mgr.blocks_manager.get_number_blocks(mgr.sequences[0]) * T > L, 'the manager maps the slot being written'
        q1 = h(r.standard_normal((B, 1, 3 * H * Dh)))
        o1 = enqueue(p_lin, q1.clone(), lin, [L, L], L, False, masked, in_len, S, smax, scales)
        o2 = enqueue(p_pg, q1.clone(), pool, [L, L], L, False, masked, in_len, S, smax, scales, pointers)
        assert torch.equal(o1, o2), f'generation step {step}'
    assert grew >= 1, 'the run must cross at least one block boundary'
    table = mgr.blocks_manager.get_pointer_array(0)
    np.testing.assert_array_equal(gather_pool(pool, table, H, T, Dh, smax), lin.cpu().numpy())


@pytest.mark.parametrize('mode,int8_kv,beam', [('fp16', 0, 1), ('sq_static_pc', 1, 1), ('fp16', 0, 3), ('woq8', 1, 2)])
@pytest.mark.parametrize('T', [8, 64])
def test_paged_session_equals_linear(mode, int8_kv, beam, T):
    """tllm_session with paged_kv_cache=1: logits after the prompt and after generation steps, and the generated tokens
    (greedy and beam search), identical to the linear-cache session."""
    cfg, w = synth_model(91)
    Bc, S, NEW = 2, 21, 9
    lens = np.array([21, 13], np.int32)
    r = np.random.default_rng(6)
    ids = np.full((Bc, S), 2, np.int32)
    for b in range(Bc):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    res = []
    for paged in (0, 1):
        s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode'], paged_kv_cache=paged, tokens_per_block=T))
        for k, v in qmodel['engine_tensors'].items():
            s.set_tensor(k, v)
        s.finalize()
        s.setup(Bc, S, NEW, beam_width=beam)
        s.context(ids, lens)
        l0 = s.logits()
        s.step(1, use_graph=False)
        l1 = s.logits()
        s.step(3, use_graph=True)
        l2 = s.logits()
        s.setup(Bc, S, NEW, beam_width=beam)
        out = s.generate(ids, lens, NEW)
        res.append((l0, l1, l2, out))
        s.close()
    for a, b in zip(res[0], res[1]):
        np.testing.assert_array_equal(a, b)


def test_bad_paged_configurations_are_rejected():
    cfg, w = synth_model(92)
    with pytest.raises(RuntimeError):
        NativeSession(dict(cfg, quant_mode=0, paged_kv_cache=1, tokens_per_block=48))
