"""CPU tests (no GPU): pin the oracle (oracle/llama_oracle.py) against the golden fixtures that were generated
from the reference / HF transformers in the build container (tests/golden/make_golden.py), and the host-side
product logic (QuantMode, weight-only quantiser) against the oracle."""
import json
import os

import numpy as np
import pytest

from oracle import llama_oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


@pytest.fixture(scope='module')
def tiny():
    return dict(np.load(os.path.join(GOLD, 'hf_tiny_llama.npz')))


def tiny_weights(t):
    w = {k: t[k].astype(np.float32) for k in ('vocab_embedding.weight', 'ln_f.weight', 'lm_head.weight')}
    w['layers'] = []
    for i in range(2):
        pre = f'layers.{i}.'
        w['layers'].append({k[len(pre):]: t[k].astype(np.float32) for k in t if k.startswith(pre)})
    return w


def test_rmsnorm_vs_hf_fp32(tiny):
    # T/tests/test_layer.py:98-135 (atol 1e-6 in fp32).  The oracle emulates fp16 io, so feed fp16-representable
    # data and compare at fp16 resolution; the fp32 formula itself is checked to 1e-6.
    x, w, y = tiny['rms_x'], tiny['rms_w'], tiny['rms_y']
    var = np.mean(x * x, -1, keepdims=True)
    np.testing.assert_allclose(x / np.sqrt(var + 1e-6) * w, y, atol=1e-6)
    x16, w16 = O.f16(x), O.f16(w)
    ref16 = x16 / np.sqrt(np.mean(x16 * x16, -1, keepdims=True) + 1e-6) * w16
    np.testing.assert_allclose(O.rmsnorm(x16, w16, 1e-6), ref16, rtol=2e-3, atol=1e-5)


def test_gated_mlp_vs_hf_fp32(tiny):
    # T/tests/test_layer.py:137-196 (atol 1e-5, and the fc<->gate_proj / gate<->up_proj naming)
    x = tiny['mlp_x'].reshape(-1, 64)
    fc, gate, proj = tiny['mlp_fc'], tiny['mlp_gate'], tiny['mlp_proj']
    inter = x @ fc.T
    y = ((inter / (1 + np.exp(-inter))) * (x @ gate.T)) @ proj.T
    np.testing.assert_allclose(y, tiny['mlp_y'].reshape(-1, 64), atol=1e-5)
    # fp16-emulating oracle on the same data
    x16 = O.f16(x)
    got = O.gemm_fp16(O.swiglu(O.gemm_fp16(x16, O.f16(fc)), O.gemm_fp16(x16, O.f16(gate))), O.f16(proj))
    np.testing.assert_allclose(got, tiny['mlp_y'].reshape(-1, 64), atol=5e-3)


def test_tiny_llama_logits_vs_hf(tiny):
    """T/tests/model/test_llama.py:286-288,352-354: logits after the context step and after one generation step,
    atol 1e-1 (fp16 engine vs HF).  The oracle is expected far inside that."""
    w = tiny_weights(tiny)
    ids, lens = tiny['ids'], tiny['input_lengths']
    B, S = ids.shape
    H, Dh, smax = 2, 32, 16
    caches = [np.zeros((B, 2, H, smax, Dh), np.float16) for _ in range(2)]
    logits = O.llama_logits_context(ids, w, caches, lens, H)
    np.testing.assert_allclose(logits, tiny['logits_ctx'], atol=1e-1)
    assert np.abs(logits - tiny['logits_ctx']).max() < 3e-2
    np.testing.assert_array_equal(logits.argmax(-1), tiny['next_ids'])
    masked = np.zeros((B, smax), np.int32)
    for b in range(B):
        masked[b, lens[b]:S] = 1
    logits2 = O.llama_logits_decode(tiny['next_ids'], w, caches, [S, S], lens, S, S, H, masked)
    np.testing.assert_allclose(logits2, tiny['logits_dec'], atol=1e-1)
    assert np.abs(logits2 - tiny['logits_dec']).max() < 3e-2


def test_quant_mode_truth_table():
    from tensorrt_llm.quantization.mode import QuantMode
    g = json.load(open(os.path.join(GOLD, 'quant_mode.json')))
    for v, preds in g['predicates'].items():
        for name, want in preds.items():
            assert bool(getattr(QuantMode(int(v)), name)()) == want, (v, name)
    for args, want in g['from_description']:
        if want is None:
            with pytest.raises(ValueError):
                QuantMode.from_description(*args)
        else:
            assert int(QuantMode.from_description(*args)) == want
    for k, want in g['use_smooth_quant'].items():
        pt, pc = [s == 'True' for s in k.split(',')]
        assert int(QuantMode.use_smooth_quant(pt, pc)) == want
    for k, want in g['use_weight_only'].items():
        assert int(QuantMode.use_weight_only(k == 'True')) == want
    assert int(QuantMode.use_smooth_quant(False, True).set_int8_kv_cache()) == g['sq_pc_int8kv'] == 46
    for n, want in g['flags'].items():
        assert int(getattr(QuantMode, n)) == want


@pytest.mark.parametrize('kind', ['dense', 'qkv'])
def test_generate_int8_vs_reference(kind):
    g = np.load(os.path.join(GOLD, 'generate_int8.npz'))
    rng = {k: g[f'{kind}_range_{k}'] for k in 'xyw'}
    got = O.generate_int8(g[f'{kind}_w'], rng, is_qkv=kind == 'qkv')
    for k, v in got.items():
        want = g[f'{kind}_out_{k}']
        if v.dtype == np.int8:
            np.testing.assert_array_equal(v, want, err_msg=k)
        else:
            assert v.dtype == want.dtype == np.float32
            np.testing.assert_allclose(v, want, rtol=2e-7, err_msg=k)


def test_smooth_gemm_vs_reference():
    g = np.load(os.path.join(GOLD, 'smooth_gemm.npz'))
    (w1, w2), s = O.smooth_gemm([g['w1'], g['w2']], g['act'], 0.5)
    # float32 pow() differs by an ulp between torch and numpy: 1e-6 relative, not bitwise
    np.testing.assert_allclose(s, g['s_joint'], rtol=1e-6)
    np.testing.assert_allclose(w1, g['w1_joint'], rtol=1e-6)
    np.testing.assert_allclose(w2, g['w2_joint'], rtol=1e-6)
    w1b, s1 = O.smooth_gemm(g['w1'], g['act'], 0.8)
    np.testing.assert_allclose(s1, g['s_single_a08'], rtol=1e-6)
    np.testing.assert_allclose(w1b, g['w1_single_a08'], rtol=1e-6)


def test_int8_rounding_edge_cases():
    # cvt.rni.sat = half-even + saturate
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5, 127.5, -128.5, 200, -200, np.nan], np.float32)
    np.testing.assert_array_equal(O.rni_sat_i8(x), [0, 2, 2, 0, -2, -2, 126, 127, -128, 127, -128, 0])


def test_kv_index_known_answers():
    # K/kvCacheUtils.h:145-169 with the KAT geometry of transposeKVKernelTest.cpp (H=8, Smax=32, Dh=256)
    H, smax, dh = 8, 32, 256
    assert O.kv_local_idx(0, 0, dh, 0, smax) == 0
    assert O.kv_local_idx(5, 3, dh, 7, smax) == 3 * smax * dh + 5 * dh + 7
    assert O.kv_flat_index(1, 1, 0, 0, 0, H, smax, dh) == 3 * H * smax * dh
    # the flat index enumerates a [B,2,H,Smax,Dh] row-major array
    a = np.arange(2 * 2 * H * smax * dh).reshape(2, 2, H, smax, dh)
    for (b, kv, h, t, c) in [(0, 0, 0, 0, 0), (1, 0, 7, 31, 255), (0, 1, 4, 16, 100), (1, 1, 2, 9, 33)]:
        assert a[b, kv, h, t, c] == O.kv_flat_index(b, kv, h, t, c, H, smax, dh)


def test_weight_only_reference_tolerance_model():
    # T/tests/quantization/_utils.py:66-88: the reference bound is 1.5 * max/2^(bits-1) = 1.5 s per column
    # (+absmax itself saturates to 2^(bits-1) - 1, a full step)
    r = np.random.default_rng(0)
    w = O.f16(r.uniform(-1, 1, (256, 64)))
    for bits in (8, 4):
        q, s = O.woq_quantize(w, bits)
        assert q.min() >= -(1 << (bits - 1)) and q.max() <= (1 << (bits - 1)) - 1
        err = np.abs(q.astype(np.float32) * s[None, :] - w)
        assert np.all(err <= 1.5 * s[None, :])
        assert np.mean(err <= 0.5 * s[None, :] * 1.01) > 0.9  # fp16-rounded scale vs fp32 divide
    # packing: low nibble first
    q4 = np.array([[1, -2, 7, -8]], np.int8)
    np.testing.assert_array_equal(O.pack_int4_kn(q4).view(np.uint8), [[0xE1, 0x87]])


def test_host_weight_quantiser_matches_oracle():
    """tllm_symmetric_quantize_last_axis (host side of the C-ABI) vs the oracle, both bit widths; and the
    processed layout documented in csrc/kernels/weight_layout.h."""
    from tensorrt_llm.plugin import capi
    r = np.random.default_rng(1)
    k, n = 96, 40
    w = r.uniform(-1, 1, (k, n)).astype(np.float16)
    for bits in (8, 4):
        processed, scales, unprocessed = capi.symmetric_quantize_last_axis(w, bits)
        q, s = O.woq_quantize(w.astype(np.float32), bits)
        np.testing.assert_array_equal(scales.astype(np.float32), s)
        if bits == 8:
            np.testing.assert_array_equal(unprocessed, q)
            assert processed.shape == (n, 96)
            np.testing.assert_array_equal(processed.view(np.uint8).astype(np.int32) - 128, q.T)
            again = capi.preprocess_weights_for_mixed_gemm(unprocessed, k, n, 8)
            np.testing.assert_array_equal(again, processed)
        else:
            np.testing.assert_array_equal(unprocessed, O.pack_int4_kn(q))
            assert processed.shape == (n, 48)
            u = processed.view(np.uint8)
            nib = np.stack([u & 0xF, u >> 4], axis=-1).reshape(n, -1)  # nibble i of each byte stream
            elem_of_nibble = [0, 2, 4, 6, 1, 3, 5, 7]
            dec = np.zeros((n, k), np.int32)
            for word in range(k // 8):
                for pos in range(8):
                    dec[:, word * 8 + elem_of_nibble[pos]] = nib[:, word * 8 + pos].astype(np.int32) - 8
            np.testing.assert_array_equal(dec, q.T)


@pytest.mark.parametrize('pos', [0, 1, 127, 1023, 2047])
def test_rope_vs_hf_rotary_at_far_positions(pos):
    """NeoX (rotate-half) RoPE of the oracle against HF's LlamaRotaryEmbedding + apply_rotary_pos_emb in fp32, at the
    positions SURVEY.md section 8c lists - position 2047 needs the angle computed in fp32 from the fp32 inverse
    frequencies exactly as HF does, or the fp16 result drifts."""
    import torch
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding, apply_rotary_pos_emb
    Dh = 128
    cfg = LlamaConfig(hidden_size=Dh * 2, num_attention_heads=2, max_position_embeddings=2048)
    rot = LlamaRotaryEmbedding(config=cfg)
    g = torch.Generator().manual_seed(pos)
    q = torch.randn(1, 2, 1, Dh, generator=g).half().float()
    cos, sin = rot(q, torch.tensor([[pos]]))
    qh, _ = apply_rotary_pos_emb(q, q, cos, sin)
    want = qh[0, :, 0].numpy()
    got = np.stack([O.apply_rope(q[0, h, 0].numpy(), pos, Dh, True) for h in range(2)])
    # oracle output is rounded to fp16 (the plugin's contract): half an fp16 ulp of |x| <= ~4
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=1e-3)
    assert np.abs(got - want).max() < 2.5e-3


@pytest.mark.parametrize('dtype', ['float16', 'float32', 'int32'])
@pytest.mark.parametrize('per_token,per_channel', [(False, False), (True, False), (False, True), (True, True)])
def test_sq_gemm_oracle_vs_the_reference_known_answer_formula(dtype, per_token, per_channel):
    """T/tests/quantization/_utils.py:91-121 `gt_matmul_smooth_quant`, restated in torch on the CPU (the original calls .cuda()):
    int32 matmul, * (scale_a x scale_b) in fp32, ROUNDED when the output type is int32, cast.  Shapes / scale distributions of
    test_smooth_quant_gemm.py:20-41,104-117 (M = 32, K = 768, N = 2304) and of test_quant_layer.py:200-303 (M = 30, K = 32, N = 64)."""
    import torch
    from oracle import llama_oracle as O
    for (m, n, k) in ((32, 2304, 768), (30, 64, 32)):
        torch.manual_seed(m + n)
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
        w = torch.randint(-128, 128, (n, k), dtype=torch.int8)
        sa = torch.ones((m if per_token else 1, 1)) * 1e-2 * torch.randint(1, 10, (m if per_token else 1, 1)).float()
        sb = torch.ones((1, n if per_channel else 1)) * 1e-2 * torch.randint(1, 10, (1, n if per_channel else 1)).float()
        ref = torch.matmul(a.to(torch.int32), w.t().to(torch.int32))
        scaling = torch.matmul(sa.expand(m, 1), sb.expand(1, n))
        ref = ref * scaling
        if dtype == 'int32':
            ref = torch.round(ref)
        ref = ref.to({'float16': torch.float16, 'float32': torch.float32, 'int32': torch.int32}[dtype])
        got = O.sq_gemm(a.numpy(), w.numpy(), sa.numpy(), sb.numpy(), dtype)
        np.testing.assert_array_equal(np.asarray(got, dtype=np.float64), ref.double().numpy())
