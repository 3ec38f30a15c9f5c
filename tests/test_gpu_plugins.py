"""GPU parity tests: every plugin of the hot path through the C-ABI (include/tllm_plugin_api.h) against the
CPU oracle (oracle/llama_oracle.py) on the same seeded inputs.  Integer paths bit-exact; fp16 paths within the
tolerances the reference's own tests state, written next to each assert.

Shapes/seeds follow the reference tests where they exist:
  T/tests/quantization/test_smooth_quant_gemm.py:20-41,104-121   (M=32, K=768, N in {2304, 3072}, scales k*1e-2)
  T/tests/quantization/test_weight_only_quant_matmul.py:112-119  ((1,1024,4096), (128,6144,12288))
  T/tests/quantization/test_functional.py:48-50,146-155          (quantisers, exact)
  T/tests/attention/test_gpt_attention.py:30-73                   (llama attention: H=4, Dh in {32,64,128})
"""
import ctypes

import numpy as np
import pytest
import torch

from helpers import HostTensor, as_f32, f32, h, i8, i32, make_plugin, run_plugin
from oracle import llama_oracle as O
from tensorrt_llm.plugin import capi

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.default_rng(seed)


# ---------------------------------------------------------------------------------------------- quantisers
@pytest.mark.parametrize('dtype', ['float16', 'float32'])
def test_quantize_tensor_exact(dtype):
    r = rng(0)
    x = r.standard_normal((4, 33, 256)).astype(np.float32) * 50
    # ties (x.5), saturation and NaN edge cases (SURVEY §8c golden list)
    x.reshape(-1)[:12] = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 127.5, -128.5, 200, -200, 126.5, np.nan]
    scale = np.float32(1.0)
    if dtype == 'float16':
        xt = h(x)
        xin = as_f32(xt)
    else:
        xt = torch.from_numpy(x).cuda()
        xin = x
    out = torch.empty(x.shape, dtype=torch.int8, device='cuda')
    p = make_plugin('QuantizeTensor', [])
    run_plugin(p, [xt, torch.tensor([[scale]], device='cuda')], [out])
    np.testing.assert_array_equal(out.cpu().numpy(), O.quantize_tensor(xin, scale))
    # reference test scale (test_functional.py:48: 0.01-ish scales)
    scale = np.float32(0.37)
    run_plugin(p, [xt, torch.tensor([[scale]], device='cuda')], [out])
    np.testing.assert_array_equal(out.cpu().numpy(), O.quantize_tensor(xin, scale))


@pytest.mark.parametrize('dtype', ['float16', 'float32'])
@pytest.mark.parametrize('shape', [(1, 4096), (7, 11008), (3, 5, 130)])
def test_quantize_per_token_exact(dtype, shape):
    r = rng(1)
    x = r.standard_normal(shape).astype(np.float32) * 3
    x[0, ..., :] *= 0  # an all-zero row exercises the 1e-6 amax floor
    if dtype == 'float16':
        xt = h(x)
        xin = as_f32(xt)
    else:
        xt = torch.from_numpy(x).cuda()
        xin = x
    q = torch.empty(shape, dtype=torch.int8, device='cuda')
    s = torch.empty(shape[:-1] + (1, ), dtype=torch.float32, device='cuda')
    run_plugin(make_plugin('QuantizePerToken', []), [xt], [q, s])
    qo, so = O.quantize_per_token(xin, is_half=dtype == 'float16')
    np.testing.assert_array_equal(s.cpu().numpy(), so)
    np.testing.assert_array_equal(q.cpu().numpy(), qo)


# ---------------------------------------------------------------------------------------------- RMSNorm
@pytest.mark.parametrize('shape', [(1, 4096), (5, 64), (2, 3, 11008)])
def test_rmsnorm(shape):
    r = rng(2)
    x = h(r.standard_normal(shape) * 2)
    g = h(1 + 0.1 * r.uniform(-1, 1, shape[-1]))
    y = torch.empty_like(x)
    p = make_plugin('Rmsnorm', [('eps', f32(1e-6)), ('type_id', i32([capi.HALF]))])
    run_plugin(p, [x, g], [y])
    ref = O.rmsnorm(as_f32(x), as_f32(g), 1e-6)
    # fp32 statistics differ only by summation order: at most 1 fp16 ulp on the output
    np.testing.assert_allclose(as_f32(y), ref, rtol=2e-3, atol=1e-5)
    assert np.mean(as_f32(y) == ref) > 0.99


@pytest.mark.parametrize('dyn', [0, 1])
def test_rmsnorm_quantization(dyn):
    r = rng(3)
    shape = (6, 4096)
    x = h(r.standard_normal(shape) * 2)
    g = h(1 + 0.1 * r.uniform(-1, 1, shape[-1]))
    scale = torch.tensor([23.5], dtype=torch.float32, device='cuda')
    q = torch.empty(shape, dtype=torch.int8, device='cuda')
    outs = [q]
    if dyn:
        s = torch.empty((6, 1), dtype=torch.float32, device='cuda')
        outs.append(s)
    p = make_plugin('RmsnormQuantization', [('eps', f32(1e-6)), ('dyn_act_scaling', i32([dyn])),
                                            ('type_id', i32([capi.HALF]))])
    run_plugin(p, [x, g, scale], outs)
    qo, so = O.rmsnorm_quant(as_f32(x), as_f32(g), 1e-6, None if dyn else 23.5)
    # +-1 LSB, the reference's own bound for LayerNorm+quant (test_smooth_quant_layer_norm.py:103-108)
    d = np.abs(q.cpu().numpy().astype(np.int32) - qo.astype(np.int32))
    assert d.max() <= 1 and np.mean(d == 0) > 0.99
    if dyn:
        np.testing.assert_allclose(s.cpu().numpy(), so, rtol=1e-2)  # test_smooth_quant_layer_norm.py:110-114


@pytest.mark.parametrize('dyn', [0, 1])
@pytest.mark.parametrize('diff_sq', [0, 1])
@pytest.mark.parametrize('shape', [(5, 768), (3, 4096), (2, 100)])
def test_layernorm_quantization(dyn, diff_sq, shape):
    """The reference's LayernormQuantization plugin (fields eps / use_diff_of_squares / dyn_act_scaling / type_id,
    inputs x, weight, bias, scale): its own test accepts +-1 LSB (test_smooth_quant_layer_norm.py:103-108)."""
    r = rng(30 + shape[1])
    x = h(r.standard_normal(shape) * 2 + 0.5)
    g = h(1 + 0.1 * r.standard_normal(shape[-1]))
    b = h(0.1 * r.standard_normal(shape[-1]))
    scale = torch.tensor([31.0], dtype=torch.float32, device='cuda')
    p = make_plugin('LayernormQuantization', [('eps', f32(1e-5)), ('use_diff_of_squares', i32([diff_sq])),
                                               ('dyn_act_scaling', i32([dyn])), ('type_id', i32([capi.HALF]))])
    q = torch.empty(shape, dtype=torch.int8, device='cuda')
    outs = [q]
    if dyn:
        s = torch.empty(shape[:-1] + (1, ), dtype=torch.float32, device='cuda')
        outs.append(s)
    run_plugin(p, [x, g, b, scale], outs)
    qo, so = O.layernorm_quant(as_f32(x), as_f32(g), as_f32(b), 1e-5, None if dyn else 31.0, bool(diff_sq))
    d = np.abs(q.cpu().numpy().astype(np.int32) - qo.astype(np.int32))
    assert d.max() <= 1 and np.mean(d == 0) > 0.98
    if dyn:
        np.testing.assert_allclose(s.cpu().numpy(), so, rtol=1e-2)
    # serialisation round trip keeps the kind
    p2 = capi.Plugin.deserialize('LayernormQuantization', p.serialize())
    assert p2 is not None and p2.plugin_type == 'LayernormQuantization' and p2.num_outputs == (2 if dyn else 1)


def test_swiglu():
    r = rng(4)
    a, b = h(r.standard_normal((3, 11008)) * 3), h(r.standard_normal((3, 11008)))
    y = torch.empty_like(a)
    run_plugin(make_plugin('SwiGLU', [('type_id', i32([capi.HALF]))]), [a, b], [y])
    np.testing.assert_allclose(as_f32(y), O.swiglu(as_f32(a), as_f32(b)), rtol=2e-3, atol=1e-4)


# ---------------------------------------------------------------------------------------------- SmoothQuant GEMM
@pytest.mark.parametrize('out_dtype', ['float16', 'float32', 'int32'])
@pytest.mark.parametrize('per_token,per_channel', [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize('m,n,k', [(32, 2304, 768), (1, 4096, 4096), (5, 3072, 768),
                                   # single-token "few rows, long K" shapes: the K-split one-shot kernel (gemv_ksplit.hip) -
                                   # the 7B down-projection, ragged row / chunk counts, its K limits (5 .. 12 KiB rows)
                                   (1, 4096, 11008), (1, 1003, 5120), (1, 13, 12288), (1, 520, 4112), (1, 64, 8192)])
def test_smooth_quant_gemm_exact(out_dtype, per_token, per_channel, m, n, k):
    torch.manual_seed(0)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
    w = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    sa = (torch.randint(1, 13, (m if per_token else 1, 1)).float() * 1e-2)  # scales k * 1e-2
    sb = (torch.randint(1, 13, (1, n if per_channel else 1)).float() * 1e-2)
    tcode = {'float16': capi.HALF, 'float32': capi.FLOAT, 'int32': capi.INT32}[out_dtype]
    tdt = {'float16': torch.float16, 'float32': torch.float32, 'int32': torch.int32}[out_dtype]
    p = make_plugin('SmoothQuantGemm', [('has_per_channel_scaling', i32(per_channel)),
                                        ('has_per_token_scaling', i32(per_token)), ('type_id', i32([tcode]))])
    out = torch.empty((m, n), dtype=tdt, device='cuda')
    run_plugin(p, [a.cuda(), w.cuda(), sa.cuda(), sb.cuda()], [out])
    ref = O.sq_gemm(a.numpy(), w.numpy(), sa.numpy(), sb.numpy(), out_dtype)
    got = out.cpu().numpy() if out_dtype == 'int32' else as_f32(out)
    np.testing.assert_array_equal(got, ref)  # exact (test_smooth_quant_gemm.py:102)


def test_smooth_quant_gemm_fp32_view_weight():
    """The reference smuggles int8 weights through an fp32 port [N, K/4] (PY/quantization/layer.py:91-99)."""
    torch.manual_seed(1)
    m, n, k = 4, 256, 512
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
    w = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    sa, sb = torch.full((1, 1), 0.03), torch.full((1, 1), 0.05)
    p = make_plugin('SmoothQuantGemm', [('has_per_channel_scaling', i32(0)), ('has_per_token_scaling', i32(0)),
                                        ('type_id', i32([capi.HALF]))])
    out = torch.empty((m, n), dtype=torch.float16, device='cuda')
    wv = w.cuda().view(torch.float32)
    assert list(wv.shape) == [n, k // 4]
    run_plugin(p, [a.cuda(), wv, sa.cuda(), sb.cuda()], [out])
    np.testing.assert_array_equal(as_f32(out), O.sq_gemm(a.numpy(), w.numpy(), sa.numpy(), sb.numpy()))


# ---------------------------------------------------------------------------------------------- weight-only
@pytest.mark.parametrize('bits', [8, 4])
@pytest.mark.parametrize('m,n,k', [(1, 1024, 4096), (3, 4096, 4096), (1, 4096, 11008), (24, 512, 1024), (1, 5120, 13824), (2, 256, 22016), (8, 4096, 11008),
                                    # prefill sizes: the fp16-expansion + LDS-DMA MFMA path (M >= 32), ragged M / N
                                    (300, 456, 1152), (64, 1024, 4096),
                                    # single token, long K: the K-split one-shot kernel with ragged row / chunk counts
                                    (1, 1000, 5120), (1, 16, 12288), (1, 520, 6144)])
def test_weight_only_quant_matmul(bits, m, n, k):
    torch.manual_seed(0)
    w = (torch.rand(k, n) * 2 - 1).half()  # [in, out], as the loaders pass it
    x = (torch.rand(m, k) * 2 - 1).half()
    processed, scales, unprocessed = capi.symmetric_quantize_last_axis(w.numpy(), bits)
    q_ref, s_ref = O.woq_quantize(w.float().numpy(), bits)
    # quantiser parity (host path): scales and integers exact
    np.testing.assert_array_equal(scales.astype(np.float32), s_ref)
    if bits == 8:
        np.testing.assert_array_equal(unprocessed, q_ref)
    else:
        np.testing.assert_array_equal(unprocessed, O.pack_int4_kn(q_ref))
    p = make_plugin('WeightOnlyQuantMatmul', [('type_id', i32([capi.HALF])), ('weight_type_id', i32(1 if bits == 8 else 2))])
    wt = torch.from_numpy(processed).cuda().view(torch.float32).reshape(k, -1)  # fp32 view [K, N/4 | N/8]
    assert wt.shape[1] == n // (4 if bits == 8 else 8)
    out = torch.empty((m, n), dtype=torch.float16, device='cuda')
    run_plugin(p, [x.cuda(), wt, torch.from_numpy(scales).cuda()], [out])
    ref = O.woq_matmul(x.float().numpy(), q_ref, s_ref)
    # reference bound: per-column atol = 1.5 * max|ref col| / 2^(bits-1) (tests/quantization/_utils.py:66-88);
    # this implementation is far inside it (fp32 accumulate of exact products):
    np.testing.assert_allclose(as_f32(out), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    # the reference's own assertion (woq_assert_colwise_near_eq, tests/quantization/_utils.py:66-88): vs the
    # dequantised-weight matmul, atol = 1.5 * max(ref col) / 2^(bits-1) (whole row when m == 1)
    deq = x.float().numpy() @ (q_ref.astype(np.float32) * s_ref[None, :])
    rs = 1.0 / (1 << (bits - 1))
    if m > 1:
        atol = np.maximum(deq.max(axis=0), 0) * rs * 1.5
        assert np.all(np.abs(as_f32(out) - deq) <= atol[None, :] + 1e-7 * np.abs(deq) + 2e-3 * np.abs(deq).max())
    else:
        np.testing.assert_allclose(as_f32(out), deq, atol=max(deq.max(), 0) * rs * 1.5)


@pytest.mark.parametrize('bits', [8, 4])
@pytest.mark.parametrize('m,n,k', [(1024, 384, 4096), (300, 456, 1152), (33, 200, 64), (257, 4096, 704)])
def test_woq_prefill_gemm_dequantises_in_the_main_loop(bits, m, n, k, lib):
    """Weight-only prefill GEMM without the fp16 image of the weights (gemm_woq.hip; reference: the mixed-input CUTLASS GEMM,
    K/cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm_template.h:60-160): every tile shape against the oracle (exact integer weights,
    fp32 accumulate, one fp16 scale, one rounding - only the fp32 summation order is the kernel's own), ragged M / N, a K of one
    and of eleven stages; and the tile shapes against each other BIT FOR BIT (the k order of a sum does not depend on the tile)."""
    lib.tllm_gemm_set_tile_cfg.argtypes = [ctypes.c_int32]
    lib.tllm_gemm_set_tile_cfg.restype = None
    r = np.random.default_rng(7 + bits)
    w = r.uniform(-1, 1, (k, n)).astype(np.float16)
    x = r.standard_normal((m, k)).astype(np.float16)
    processed, scales, _ = capi.symmetric_quantize_last_axis(w, bits)
    q_ref, s_ref = O.woq_quantize(w.astype(np.float32), bits)
    ref = O.woq_matmul(x.astype(np.float32), q_ref, s_ref)
    p = make_plugin('WeightOnlyQuantMatmul', [('type_id', i32([capi.HALF])), ('weight_type_id', i32(1 if bits == 8 else 2))])
    wt = torch.from_numpy(processed).cuda().view(torch.float32).reshape(k, -1)
    outs = []
    try:
        for cfg in (101, 102, 103, 104, 105, 106):
            lib.tllm_gemm_set_tile_cfg(cfg)
            out = torch.full((m, n), 7.0, dtype=torch.float16, device='cuda')
            run_plugin(p, [torch.from_numpy(x).cuda(), wt, torch.from_numpy(scales).cuda()], [out])
            got = as_f32(out)
            np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max(), err_msg=f'cfg {cfg}')
            outs.append(got)
    finally:
        lib.tllm_gemm_set_tile_cfg(0)
    for g in outs[1:]:
        np.testing.assert_array_equal(g, outs[0])


@pytest.mark.parametrize('bits', [8, 4])
@pytest.mark.parametrize('m,n,k', [(1, 4096, 11008), (2, 4096, 11008), (1, 12288, 4096), (4, 4096, 4096)])
def test_weight_only_gemv_with_a_dc_offset_in_the_activations(bits, m, n, k):
    """the weight-only GEMV runs its dot products on the raw byte / nibble splices (1152 + q, 1024 + n,
    1024 + 16 n) and takes the splice bias times sum(x) off afterwards.  With zero-mean activations that bias term is small; with
    a DC offset (x = 3 + N(0, 1): SwiGLU outputs into mlp.proj, attention outputs) it is 16x (int8) to 220x (int4) the signal and
    fp32 rounding must still cancel.  Checked against the exact result (float64 sum of exact products, one rounding), in units of
    the fp16 spacing at the largest output of the row."""
    r = np.random.default_rng(3)
    w = (r.uniform(-1, 1, (k, n))).astype(np.float16)
    x = (3.0 + r.standard_normal((m, k))).astype(np.float16)
    processed, scales, _ = capi.symmetric_quantize_last_axis(w, bits)
    q_ref, s_ref = O.woq_quantize(w.astype(np.float32), bits)
    p = make_plugin('WeightOnlyQuantMatmul', [('type_id', i32([capi.HALF])), ('weight_type_id', i32(1 if bits == 8 else 2))])
    wt = torch.from_numpy(processed).cuda().view(torch.float32).reshape(k, -1)
    out = torch.empty((m, n), dtype=torch.float16, device='cuda')
    run_plugin(p, [torch.from_numpy(x).cuda(), wt, torch.from_numpy(scales).cuda()], [out])
    exact = (x.astype(np.float64) @ q_ref.astype(np.float64)) * s_ref.astype(np.float64)[None, :]
    got = as_f32(out).astype(np.float64)
    top = np.abs(exact).max()
    ulp = 2.0 ** (np.floor(np.log2(top)) - 10)  # fp16 spacing at the largest output
    err = np.abs(got - exact)
    print(f'[woq{bits} m={m} n={n} k={k}] max |err| = {err.max() / ulp:.2f} fp16 ulp(max |y|), rms = {np.sqrt((err ** 2).mean()) / ulp:.3f}, '
          f'|sum x| = {np.abs(x.astype(np.float64).sum(1)).max():.0f}, max |y| = {top:.1f}')
    # one rounding of the exact value is <= 0.5 ulp; fp32 accumulation of the splices must not add more than another 0.5
    assert err.max() <= 1.0 * ulp, err.max() / ulp


@pytest.mark.parametrize('m,n,k', [(1, 4096, 4096), (2, 32000, 4096), (1, 4096, 11008), (19, 192, 64), (8, 24, 64),
                                   (1, 1003, 5120), (1, 13, 12288), (1, 520, 2056),  # K-split kernel, ragged rows / chunks
                                   # rows beyond 12288 halfs: the down-projections of LLaMA-13B / 65B (the third activation bucket, r04)
                                   (1, 5120, 13824), (3, 512, 22016),
                                   # 8 rows of an fp16 K = 11008 vector exceed a CU's LDS: slabs of 4 rows (r04; refused before)
                                   (8, 4096, 11008), (5, 256, 13824)])
def test_gemm_fp16(m, n, k):
    r = rng(5)
    x, w = h(r.standard_normal((m, k))), h(r.standard_normal((n, k)) / np.sqrt(k))
    p = make_plugin('Gemm', [('transa', i32(0)), ('transb', i32(1)), ('type_id', i32([capi.HALF]))])
    out = torch.empty((m, n), dtype=torch.float16, device='cuda')
    run_plugin(p, [x, w], [out])
    np.testing.assert_allclose(as_f32(out), O.gemm_fp16(as_f32(x), as_f32(w)), rtol=2e-3, atol=2e-3)


# ---------------------------------------------------------------------------------------------- prefill GEMM tiles
@pytest.mark.parametrize('cfg', list(range(1, 13)) + [15, 18, 20, 36, 37, 42, 50, 51, 52, 53, 54, 55, 56, 60, 62, 63])
def test_prefill_gemm_every_tile_shape(cfg):
    """Every tile shape of the LDS-DMA staged MFMA GEMM (kernels/gemm_glds.hip; tllm_gemm_set_tile_cfg) on a problem
    with ragged M / N edges and several K-tiles: SmoothQuant exact (int32 accumulation is order-independent, the
    epilogue is float(acc) * (s_col * s_row) -> fp16), fp16 within the fp16 GEMM tolerance."""
    lib = capi.load_library()
    lib.tllm_gemm_set_tile_cfg.argtypes = [__import__('ctypes').c_int32]
    lib.tllm_gemm_set_tile_cfg.restype = None
    lib.tllm_gemm_set_tile_cfg(cfg)
    try:
        torch.manual_seed(cfg)
        m, n, k = 300, 456, 1152
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
        w = torch.randint(-128, 128, (n, k), dtype=torch.int8)
        sa = (torch.randint(1, 13, (m, 1)).float() * 1e-2)
        sb = (torch.randint(1, 13, (1, n)).float() * 1e-2)
        for out_dtype, tcode, tdt in (('float16', capi.HALF, torch.float16), ('int32', capi.INT32, torch.int32)):
            p = make_plugin('SmoothQuantGemm', [('has_per_channel_scaling', i32(1)), ('has_per_token_scaling', i32(1)),
                                                ('type_id', i32([tcode]))])
            out = torch.empty((m, n), dtype=tdt, device='cuda')
            run_plugin(p, [a.cuda(), w.cuda(), sa.cuda(), sb.cuda()], [out])
            ref = O.sq_gemm(a.numpy(), w.numpy(), sa.numpy(), sb.numpy(), out_dtype)
            got = out.cpu().numpy() if out_dtype == 'int32' else as_f32(out)
            np.testing.assert_array_equal(got, ref)
        r = rng(40 + cfg)
        x, wf = h(r.standard_normal((m, 320))), h(r.standard_normal((n, 320)) / np.sqrt(320))
        p = make_plugin('Gemm', [('transa', i32(0)), ('transb', i32(1)), ('type_id', i32([capi.HALF]))])
        out = torch.empty((m, n), dtype=torch.float16, device='cuda')
        run_plugin(p, [x, wf], [out])
        np.testing.assert_allclose(as_f32(out), O.gemm_fp16(as_f32(x), as_f32(wf)), rtol=2e-3, atol=2e-3)
    finally:
        lib.tllm_gemm_set_tile_cfg(0)


# ---------------------------------------------------------------------------------------------- attention
def attention_plugin(H, Dh, int8_kv, rot=None, neox=1, packed=0, q_scaling=1.0):
    return make_plugin('GPTAttention', [
        ('num_heads', i32(H)), ('head_size', i32(Dh)), ('unidirectional', i32(1)), ('q_scaling', f32(q_scaling)),
        ('rotary_embedding_dim', i32(Dh if rot is None else rot)), ('neox_rotary_style', i8(neox)),
        ('context_fmha_type', i8(0)), ('multi_block_mode', i8(0)), ('multi_query_mode', i8(0)),
        ('int8_kv_cache', i32(int8_kv)), ('fp8_kv_cache', i32(0)), ('remove_input_padding', i8(packed)),
        ('mask_type', i32([1])), ('paged_kv_cache', i32(0)), ('type_id', i32([capi.HALF])), ('in_flight_batching', i32(0)),
    ])


def run_attention(p, qkv, cache, seq_len, past_len, is_context, masked, in_len, max_in, smax, scales=None,
                  cache_indirection=None):
    B = qkv.shape[0]
    out = torch.empty(qkv.shape[:-1] + (qkv.shape[-1] // 3, ), dtype=torch.float16, device='cuda')
    ins = [qkv, cache, torch.tensor(seq_len, dtype=torch.int32, device='cuda'),
           HostTensor([past_len, 1 if is_context else 0]),
           torch.tensor(masked, dtype=torch.int32, device='cuda'),
           torch.tensor(in_len, dtype=torch.int32, device='cuda'),
           HostTensor(np.zeros(0), shape=[max_in]),  # value carried by the shape
           HostTensor(np.zeros(0), shape=[B, 1, smax])]
    # shape-only tensors still need a non-null device pointer in the reference; give them one
    dummy = torch.zeros(max(max_in, B * smax), dtype=torch.int32, device='cuda')
    ins[6] = dummy[:max_in]
    ins[7] = dummy[:B * smax].view(B, 1, smax)
    if cache_indirection is not None:  # int32 [batch, beam_width, smax]
        ins[7] = torch.from_numpy(np.ascontiguousarray(cache_indirection, dtype=np.int32)).cuda()
    if scales is not None:
        ins += [torch.tensor([scales[0]], dtype=torch.float32, device='cuda'),
                torch.tensor([scales[1]], dtype=torch.float32, device='cuda')]
    run_plugin(p, ins, [out, cache])
    return out


@pytest.mark.parametrize('int8_kv', [0, 1])
@pytest.mark.parametrize('H,Dh', [(4, 128), (4, 64), (2, 32), (32, 128)])
@pytest.mark.parametrize('L', [1, 17, 128, 1023, 2047, 4000])
def test_mmha_decode_vs_oracle(int8_kv, H, Dh, L):
    """Generation step: new token at slot L; half of batch element 1's prompt is padding (test_gpt_attention.py:437-449)."""
    r = rng(100 + L)
    B, smax = 2, (1152 if L < 1152 else (2048 if L < 2048 else 4096))  # L = 2047: the last slot of a full n_positions = 2048 cache
    max_in = max(L - 3, 1) if L > 4 else L
    in_len = [max_in, max(max_in // 2, 1)]
    masked = np.zeros((B, smax), dtype=np.int32)
    masked[1, in_len[1]:max_in] = 1
    kv_scale = 0.05  # int8 scale: values ~N(0,1) -> +-6 sigma covered by 127 * 0.05
    scales = (1.0 / kv_scale, kv_scale) if int8_kv else None
    past = r.standard_normal((B, 2, H, smax, Dh)).astype(np.float32)
    past[:, :, :, L:] = 0
    if int8_kv:
        cache_np = O.rni_sat_i8(past * np.float32(scales[0]))
        cache = torch.from_numpy(cache_np.copy()).cuda()
    else:
        cache_np = past.astype(np.float16)
        cache = torch.from_numpy(cache_np.copy()).cuda()
    qkv = h(r.standard_normal((B, 1, 3 * H * Dh)))
    p = attention_plugin(H, Dh, int8_kv)
    out = run_attention(p, qkv, cache, [L, L], L, False, masked, in_len, max_in, smax, scales)
    ref_cache = cache_np.copy()
    ref = O.mmha_decode(as_f32(qkv)[:, 0], ref_cache, [L, L], in_len, max_in, L, H, Dh, Dh, True, 1.0, masked,
                        scales[0] if scales else None, scales[1] if scales else None)
    # generation-step tolerance of the reference's plugin test: atol 2e-3 (test_gpt_attention.py:828-831)
    np.testing.assert_allclose(as_f32(out)[:, 0], ref, atol=2e-3, rtol=0)
    # cache: nothing but slot L may change; slot L within 1 fp16 ulp / 1 int8 LSB of the oracle (RoPE uses fma)
    got = cache.cpu().numpy()
    keep = np.ones(smax, dtype=bool)
    keep[L] = False
    np.testing.assert_array_equal(got[:, :, :, keep], cache_np[:, :, :, keep])
    if int8_kv:
        d = np.abs(got[:, :, :, L].astype(np.int32) - ref_cache[:, :, :, L].astype(np.int32))
        assert d.max() <= 1 and np.mean(d == 0) > 0.98
    else:
        np.testing.assert_allclose(got[:, :, :, L].astype(np.float32), ref_cache[:, :, :, L].astype(np.float32),
                                   atol=2e-4, rtol=2e-3)  # KV tolerance of test_gpt_attention.py:561-578 (+1 ulp)


@pytest.mark.parametrize('int8_kv', [0, 1])
@pytest.mark.parametrize('H,Dh,L,W', [(4, 128, 200, 3), (2, 64, 37, 2), (4, 128, 1023, 4)])
def test_mmha_decode_beam_cache_indirection(int8_kv, H, Dh, L, W):
    """Beam search generation step (P/gptAttentionPlugin/gptAttentionPlugin.cpp:330-336, MM/...Template.h:1137-1146,
    1624-1631): sequence b * W + k reads time step t from the cache rows of sequence b * W + cache_indirection[b, k, t];
    the new token's K/V go to the sequence's own rows."""
    r = rng(300 + L)
    batch, smax = 2, 1152
    B = batch * W
    max_in = L - 3
    in_len = np.repeat([max_in, max_in // 2], W).tolist()
    masked = np.zeros((B, smax), dtype=np.int32)
    masked[W:, in_len[W]:max_in] = 1
    kv_scale = 0.05
    scales = (1.0 / kv_scale, kv_scale) if int8_kv else None
    past = r.standard_normal((B, 2, H, smax, Dh)).astype(np.float32)
    past[:, :, :, L:] = 0
    cache_np = O.rni_sat_i8(past * np.float32(scales[0])) if int8_kv else past.astype(np.float16)
    cache = torch.from_numpy(cache_np.copy()).cuda()
    ci = r.integers(0, W, (batch, W, smax)).astype(np.int32)
    qkv = h(r.standard_normal((B, 1, 3 * H * Dh)))
    p = attention_plugin(H, Dh, int8_kv)
    out = run_attention(p, qkv, cache, [L] * B, L, False, masked, in_len, max_in, smax, scales, cache_indirection=ci)
    # the oracle sees, per sequence, the cache its indirection row selects
    eff = np.empty_like(cache_np)
    for bb in range(B):
        b, k = divmod(bb, W)
        src = b * W + ci[b, k]  # [smax]
        eff[bb] = cache_np[src, :, :, np.arange(smax)].transpose(1, 2, 0, 3)
    ref_cache = eff.copy()
    ref = O.mmha_decode(as_f32(qkv)[:, 0], ref_cache, [L] * B, in_len, max_in, L, H, Dh, Dh, True, 1.0, masked,
                        scales[0] if scales else None, scales[1] if scales else None)
    np.testing.assert_allclose(as_f32(out)[:, 0], ref, atol=2e-3, rtol=0)
    got = cache.cpu().numpy()
    keep = np.ones(smax, dtype=bool)
    keep[L] = False
    np.testing.assert_array_equal(got[:, :, :, keep], cache_np[:, :, :, keep])  # siblings' rows untouched
    if int8_kv:
        d = np.abs(got[:, :, :, L].astype(np.int32) - ref_cache[:, :, :, L].astype(np.int32))
        assert d.max() <= 1 and np.mean(d == 0) > 0.98
    else:
        np.testing.assert_allclose(got[:, :, :, L].astype(np.float32), ref_cache[:, :, :, L].astype(np.float32),
                                   atol=2e-4, rtol=2e-3)


@pytest.mark.parametrize('int8_kv', [0, 1])
@pytest.mark.parametrize('Dh,rot,q_scaling', [(128, 128, 1.0), (128, 64, 2.0), (64, 32, 0.5), (32, 32, 1.0)])
def test_gptj_rotary_partial_dim_and_q_scaling(int8_kv, Dh, rot, q_scaling):
    """The plugin fields LLaMA leaves at their defaults: GPT-J style RoPE (pairs (2i, 2i+1), neox_rotary_style = 0), a
    rotary dimension smaller than the head (the tail is not rotated) and q_scaling != 1 (inv_sqrt_dh = 1 / (sqrt(Dh) *
    q_scaling), gptAttentionCommon.cpp:163) - context phase and one generation step against the oracle."""
    r = rng(500 + Dh + rot)
    H, B, S = 4, 2, 70
    smax = S + 8
    in_len = [S, S - 9]
    kv_scale = 0.05
    scales = (1.0 / kv_scale, kv_scale) if int8_kv else None
    p = attention_plugin(H, Dh, int8_kv, rot=rot, neox=0, q_scaling=q_scaling)
    qkv = h(r.standard_normal((B, S, 3 * H * Dh)))
    qkv_in = qkv.clone()
    dt = torch.int8 if int8_kv else torch.float16
    cache = torch.zeros((B, 2, H, smax, Dh), dtype=dt, device='cuda')
    masked = np.zeros((B, smax), np.int32)
    masked[1, in_len[1]:S] = 1
    out = run_attention(p, qkv, cache, [S, S], 0, True, masked, in_len, S, smax, scales)
    ref_cache = np.zeros((B, 2, H, smax, Dh), dtype=np.int8 if int8_kv else np.float16)
    ref, _ = O.context_attention(as_f32(qkv_in), ref_cache, in_len, H, Dh, rot, False, q_scaling, scales[0] if scales else None)
    np.testing.assert_allclose(as_f32(out), ref, atol=5e-3, rtol=0)
    got = cache.cpu().numpy()
    np.testing.assert_array_equal(got[:, 1], ref_cache[:, 1])  # V is never rotated
    if not int8_kv:
        np.testing.assert_allclose(got[:, 0].astype(np.float32), ref_cache[:, 0].astype(np.float32), atol=2e-4, rtol=2e-3)
        # the un-rotated tail of K is a pure copy
        src = as_f32(qkv_in).reshape(B, S, 3, H, Dh)
        for b in range(B):
            np.testing.assert_array_equal(got[b, 0, :, :in_len[b], rot:].astype(np.float32),
                                          src[b, :in_len[b], 1, :, rot:].transpose(1, 0, 2))
    # one generation step on top of the oracle's cache
    cache = torch.from_numpy(ref_cache.copy()).cuda()
    q1 = h(r.standard_normal((B, 1, 3 * H * Dh)))
    out = run_attention(p, q1, cache, [S, S], S, False, masked, in_len, S, smax, scales)
    ref = O.mmha_decode(as_f32(q1)[:, 0], ref_cache, [S, S], in_len, S, S, H, Dh, rot, False, q_scaling, masked,
                        scales[0] if scales else None, scales[1] if scales else None)
    np.testing.assert_allclose(as_f32(out)[:, 0], ref, atol=2e-3, rtol=0)


@pytest.mark.parametrize('int8_kv', [0, 1])
def test_kv_cache_append_bit_exact(int8_kv):
    """A3 bit-exact row: without RoPE the appended K/V are pure copies / quantisations, so the cache must equal
    the oracle's kv_flat_index/kv_store element for element (layout K/kvCacheUtils.h:114-170; KAT shape of
    T/cpp/tests/runtime/transposeKVKernelTest.cpp: B=2, H=8, Dh=256, Smax=32, scale 0.1)."""
    r = rng(7)
    B, H, Dh, smax, L = 2, 8, 256, 32, 16
    scale = 0.1
    x = (r.uniform(-1, 1, (B, 1, 3 * H * Dh)) / scale).astype(np.float32)
    x.reshape(-1)[:8] = np.array([0.5, 1.5, 2.5, -0.5, -1.5, 1270, -1290, 3.5]) / scale  # ties + saturation
    qkv = h(x)
    dt = torch.int8 if int8_kv else torch.float16
    cache = torch.zeros((B, 2, H, smax, Dh), dtype=dt, device='cuda')
    p = attention_plugin(H, Dh, int8_kv, rot=0)
    scales = (scale, 1.0 / scale) if int8_kv else None
    run_attention(p, qkv, cache, [L, L], L, False, np.zeros((B, smax), np.int32), [L, L], L, smax, scales)
    flat = cache.cpu().numpy().reshape(-1)
    src = as_f32(qkv).reshape(B, 3, H, Dh)
    expect = np.zeros_like(flat)
    for b in range(B):
        for kv in range(2):
            for hh in range(H):
                vals = O.kv_store(src[b, 1 + kv, hh], scale if int8_kv else None)
                i0 = O.kv_flat_index(b, kv, hh, L, 0, H, smax, Dh)
                expect[i0:i0 + Dh] = vals
    np.testing.assert_array_equal(flat, expect)


@pytest.mark.parametrize('int8_kv', [0, 1])
@pytest.mark.parametrize('H,Dh,S', [(4, 128, 90), (2, 32, 128), (4, 64, 33), (4, 128, 300), (2, 64, 257), (2, 128, 1024),
                                    # block boundaries of the MFMA kernel (64-key blocks, 128-query workgroups) and n_positions
                                    (2, 128, 64), (2, 128, 65), (2, 64, 127), (2, 128, 129), (1, 128, 2048),
                                    # beyond LLaMA-1's 2048 positions (4096-position checkpoints): 47 / 64 key blocks per head
                                    (2, 128, 3000), (1, 64, 4096),
                                    # >= 256 workgroups: the 8-wave kernel with paired 64-query blocks (16 blocks; 15 blocks: the
                                    # middle one has no partner; 13 blocks with a ragged last one)
                                    (16, 128, 1024), (16, 128, 960), (32, 128, 400), (32, 128, 1024)])
def test_context_attention_vs_oracle(int8_kv, H, Dh, S):
    r = rng(200 + S)
    B, smax = 2, S + 8
    in_len = [S, S // 2]  # half of sequence 1 is padding
    qkv = h(r.standard_normal((B, S, 3 * H * Dh)))
    dt = torch.int8 if int8_kv else torch.float16
    cache = torch.zeros((B, 2, H, smax, Dh), dtype=dt, device='cuda')
    kv_scale = 0.05
    scales = (1.0 / kv_scale, kv_scale) if int8_kv else None
    p = attention_plugin(H, Dh, int8_kv)
    qkv_in = qkv.clone()
    masked = np.zeros((B, smax), np.int32)
    out = run_attention(p, qkv, cache, [S, S], 0, True, masked, in_len, S, smax, scales)
    ref_cache = np.zeros((B, 2, H, smax, Dh), dtype=np.int8 if int8_kv else np.float16)
    ref, _ = O.context_attention(as_f32(qkv_in), ref_cache, in_len, H, Dh, Dh, True, 1.0, scales[0] if scales else None)
    # context tolerance of the reference's plugin test: atol 5e-3 (test_gpt_attention.py:685-695)
    np.testing.assert_allclose(as_f32(out), ref, atol=5e-3, rtol=0)
    got = cache.cpu().numpy()
    if int8_kv:
        d = np.abs(got.astype(np.int32) - ref_cache.astype(np.int32))
        assert d.max() <= 1 and np.mean(d == 0) > 0.98
        # V is never rotated: bit-exact
        np.testing.assert_array_equal(got[:, 1], ref_cache[:, 1])
    else:
        np.testing.assert_array_equal(got[:, 1], ref_cache[:, 1])
        # rotated keys: the reference's KV tolerance (test_gpt_attention.py:561-578).  Beyond 2048 positions the fp32 ANGLE itself
        # (pos / 10000^(2j/rot), up to 4095 rad) is only defined to one ulp = 2.4e-4 rad - powf of two libraries differs by that -
        # so cos / sin, and with them a rotated key of magnitude ~1, move by as much
        np.testing.assert_allclose(got[:, 0].astype(np.float32), ref_cache[:, 0].astype(np.float32), atol=2e-4 if S <= 2048 else 8e-4,
                                   rtol=2e-3)


@pytest.mark.parametrize('int8_kv', [0, 1])
@pytest.mark.parametrize('H,Dh,S,env', [(4, 128, 200, ''), (2, 64, 70, ''), (2, 32, 40, ''), (32, 128, 1000, ''),
                                        # shapes at which the launcher picks its other kernel variants by workgroup count (with
                                        # 3 sequences): paired 64-query blocks at 32 heads x 300 tokens, 64-query workgroups at 4 heads
                                        (32, 128, 300, 'oracle'), (32, 64, 333, 'oracle'), (4, 128, 333, 'oracle')])
def test_packed_context_attention_equals_padded(int8_kv, H, Dh, S, env, monkeypatch):
    """remove_input_padding (gptAttentionPlugin.cpp:344-356): the tokens of all sequences back to back in [1, T, 3 D];
    the plugin must produce, for every real token, what the padded run produces, and the same KV cache.  The rows marked
    'oracle' are also held against the oracle (the padded run; 5e-3 as in test_context_attention_vs_oracle)."""
    r = rng(300 + S)
    B, smax = 3, S + 8
    in_len = [S, S // 3, max(S // 2, 1)]
    qkv = h(r.standard_normal((B, S, 3 * H * Dh)))
    dt = torch.int8 if int8_kv else torch.float16
    scales = (20.0, 0.05) if int8_kv else None
    masked = np.zeros((B, smax), np.int32)
    cache_pad = torch.zeros((B, 2, H, smax, Dh), dtype=dt, device='cuda')
    out_pad = run_attention(attention_plugin(H, Dh, int8_kv), qkv.clone(), cache_pad, [S] * B, 0, True, masked, in_len, S, smax,
                            scales)
    if env:
        ref_cache = np.zeros((B, 2, H, smax, Dh), dtype=np.int8 if int8_kv else np.float16)
        ref, _ = O.context_attention(as_f32(qkv), ref_cache, in_len, H, Dh, Dh, True, 1.0, scales[0] if scales else None)
        np.testing.assert_allclose(as_f32(out_pad), ref, atol=5e-3, rtol=0)
    packed = torch.cat([qkv[b, :in_len[b]] for b in range(B)], dim=0)[None].contiguous()  # [1, T, 3 D]
    cache_pk = torch.zeros((B, 2, H, smax, Dh), dtype=dt, device='cuda')
    p = attention_plugin(H, Dh, int8_kv, packed=1)
    T = packed.shape[1]
    out_pk = torch.empty((1, T, H * Dh), dtype=torch.float16, device='cuda')
    ins = [packed, cache_pk, torch.tensor([S] * B, dtype=torch.int32, device='cuda'), HostTensor([0, 1]),
           torch.tensor(masked, dtype=torch.int32, device='cuda'), torch.tensor(in_len, dtype=torch.int32, device='cuda')]
    dummy = torch.zeros(max(S, B * smax), dtype=torch.int32, device='cuda')
    ins += [dummy[:S], dummy[:B * smax].view(B, 1, smax)]
    if scales:
        ins += [torch.tensor([scales[0]], dtype=torch.float32, device='cuda'),
                torch.tensor([scales[1]], dtype=torch.float32, device='cuda')]
    run_plugin(p, ins, [out_pk, cache_pk])
    off = 0
    for b in range(B):
        # same kernels, same per-row arithmetic: bit-identical rows
        assert torch.equal(out_pk[0, off:off + in_len[b]], out_pad[b, :in_len[b]]), b
        off += in_len[b]
    assert torch.equal(cache_pk, cache_pad)


def test_plugin_rejects_unbuilt_features():
    p = capi.Plugin.create('GPTAttention', [capi.PluginField('num_heads', i32(4))])
    assert p is None and 'missing plugin field' in capi.last_error()


# ---------------------------------------------------------------------------------------------- collectives
def test_allreduce_allgather_single_rank_rccl():
    """The comm plugins on a 1-rank RCCL communicator (the only world a 1-GPU box allows): exercises the dlopen of
    librccl, ncclCommInitRank through tllm_comm_init_rank, the communicator lookup by rank set
    (P/common/plugin.h:181-188) and the enqueue path of P/ncclPlugin/{allreduce,allgather}Plugin.cpp; with one
    rank both collectives are the identity.  N > 1 is covered on CPU by tests/test_tp_gloo.py."""
    import ctypes
    lib = capi.load_library()
    uid = (ctypes.c_char * 128)()
    assert lib.tllm_comm_get_unique_id(uid) == 0, capi.last_error()
    group = (ctypes.c_int32 * 1)(0)
    assert lib.tllm_comm_init_rank(group, 1, 0, uid) == 0, capi.last_error()
    try:
        r = rng(21)
        x = h(r.standard_normal((3, 4096)))
        y = torch.zeros_like(x)
        ar = make_plugin('AllReduce', [('group', i32([0])), ('type_id', i32([capi.HALF]))])
        run_plugin(ar, [x], [y])
        assert torch.equal(x, y)
        ag = make_plugin('AllGather', [('group', i32([0])), ('type_id', i32([capi.HALF]))])
        z = torch.zeros_like(x)
        run_plugin(ag, [x], [z])
        assert torch.equal(x, z)
        # a group nobody initialised is an error, not a hang or an exit()
        bad = make_plugin('AllReduce', [('group', i32([0, 1])), ('type_id', i32([capi.HALF]))])
        with pytest.raises(RuntimeError):
            run_plugin(bad, [x], [y])
    finally:
        assert lib.tllm_comm_destroy_all() == 0


# ---------------------------------------------------------------------------------------------- fused MLP GEMM
@pytest.mark.parametrize('m,n,k', [(300, 456, 1152), (1024, 1376, 512), (64, 96, 128), (257, 200, 256),
                                   # several tiles per persistent workgroup (r05): 460 / 440 tiles on 256 CUs, odd K-tile count, ragged rows
                                   (1024, 11008, 384), (1000, 10560, 640)])
def test_dual_gemm_swiglu_quant_equals_the_unfused_chain(m, n, k):
    """tllm_gemm_swiglu_quant (fc and gate of the SmoothQuant MLP in one kernel, SwiGLU + static quantiser in its epilogue,
    kernels/gemm_sqp.hip DUAL) against the chain it replaces in the prefill - two exact SmoothQuant GEMMs to fp16
    (tests above), SwiGLU with the reference graph's fp16 rounding points (PY/layers/mlp.py:68-73) and the static int8
    quantiser: every int8 must be identical.  Ragged M / N edges, several K-tiles, N not a multiple of 16 (scalar stores)."""
    import ctypes
    lib = capi.load_library()

    class GemmParams(ctypes.Structure):
        _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                    ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                    ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                    ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p),
                    ('ldc', ctypes.c_int64)]

    lib.tllm_gemm_swiglu_quant.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.tllm_gemm_swiglu_quant.restype = ctypes.c_int32
    torch.manual_seed(m + n)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
    w1 = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    w2 = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    s1 = (torch.randint(1, 13, (n, )).float() * 2e-5)
    s2 = (torch.randint(1, 13, (n, )).float() * 2e-5)
    sr = torch.tensor([0.75], dtype=torch.float32)
    qs = torch.tensor([23.0], dtype=torch.float32)
    # the un-fused chain on the CPU with the oracle's rounding points
    acc1 = (a.double() @ w1.double().t()).float()
    acc2 = (a.double() @ w2.double().t()).float()
    g16 = (acc1 * (s1[None, :] * sr)).half().float()
    u16 = (acc2 * (s2[None, :] * sr)).half().float()
    a16 = (g16 / (1.0 + torch.exp(-g16))).half().float()
    o16 = (a16 * u16).half().float()
    ref = torch.clamp(torch.round(o16 * qs), -128, 127).to(torch.int8)
    d = {k_: v.cuda() for k_, v in dict(a=a, w1=w1, w2=w2, s1=s1, s2=s2, sr=sr, qs=qs).items()}
    out = torch.full((m, n), 77, dtype=torch.int8, device='cuda')
    q = GemmParams(3, 2, m, n, k, d['a'].data_ptr(), k, d['w1'].data_ptr(), k, d['s1'].data_ptr(), d['sr'].data_ptr(), 1, 0,
                   out.data_ptr(), n)
    rc = lib.tllm_gemm_swiglu_quant(ctypes.byref(q), d['w2'].data_ptr(), d['s2'].data_ptr(), d['qs'].data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
    assert rc == 0, capi.last_error()
    torch.cuda.synchronize()
    got = out.cpu()
    # exp() on the GPU (v_exp_f32) and in torch differ in the last ulp: a result that sits on an fp16 / int8 rounding boundary may
    # flip by one LSB - allow a handful, never more than 1
    diff = (got.int() - ref.int()).abs()
    assert int(diff.max()) <= 1 and int((diff > 0).sum()) <= max(4, m * n // 2000), (int(diff.max()), int((diff > 0).sum()))
    # the persistent form (int8 rows on 16-byte boundaries) and the one-tile-per-workgroup form: the same bytes
    lib.tllm_gemm_set_tile_cfg.argtypes = [ctypes.c_int32]
    lib.tllm_gemm_set_tile_cfg.restype = None
    lib.tllm_gemm_set_tile_cfg(-2)
    try:
        out2 = torch.full((m, n), 55, dtype=torch.int8, device='cuda')
        q.c = out2.data_ptr()
        assert lib.tllm_gemm_swiglu_quant(ctypes.byref(q), d['w2'].data_ptr(), d['s2'].data_ptr(), d['qs'].data_ptr(),
                                          torch.cuda.current_stream().cuda_stream) == 0, capi.last_error()
        torch.cuda.synchronize()
        assert torch.equal(out2.cpu(), got)
    finally:
        lib.tllm_gemm_set_tile_cfg(0)


def test_gemm_clock_probe_reports_a_plausible_shader_clock():
    """tllm_gemm_set_clock_probe ("the chip clocks to its power budget"): while set, every workgroup of
    the SmoothQuant prefill GEMMs writes {shader cycles, ticks of the constant 100 MHz counter}; the ratio is the clock the
    chip held - between a deep-throttle floor and the 2.4 GHz maximum - and the GEMM's result is unaffected.  Unset: nothing
    is written."""
    import ctypes
    lib = capi.load_library()
    lib.tllm_gemm_set_clock_probe.argtypes = [ctypes.c_void_p]
    lib.tllm_gemm_set_clock_probe.restype = None
    torch.manual_seed(3)
    m, n, k = 1024, 1536, 2048   # 256 x 192 tiles: the phased kernel; then 128 x 128: the lock-step one
    for n_ in (n, 512):
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
        w = torch.randint(-128, 128, (n_, k), dtype=torch.int8)
        sa = (torch.randint(1, 13, (m, 1)).float() * 1e-2)
        sb = (torch.randint(1, 13, (1, n_)).float() * 1e-2)
        p = make_plugin('SmoothQuantGemm', [('has_per_channel_scaling', i32(1)), ('has_per_token_scaling', i32(1)),
                                            ('type_id', i32([capi.HALF]))])
        ref = O.sq_gemm(a.numpy(), w.numpy(), sa.numpy(), sb.numpy(), 'float16')
        probe = torch.zeros(8192, dtype=torch.int64, device='cuda')
        out = torch.empty((m, n_), dtype=torch.float16, device='cuda')
        try:
            lib.tllm_gemm_set_clock_probe(ctypes.c_void_p(probe.data_ptr()))
            for _ in range(4):
                run_plugin(p, [a.cuda(), w.cuda(), sa.cuda(), sb.cuda()], [out])
            torch.cuda.synchronize()
        finally:
            lib.tllm_gemm_set_clock_probe(None)
        np.testing.assert_array_equal(as_f32(out), ref)
        d = probe.view(-1, 2).cpu().numpy()
        used = d[d[:, 1] > 0]
        assert len(used) >= 8, 'no workgroup reported'
        mhz = np.median(used[:, 0] / used[:, 1]) * 100.0
        print(f'N = {n_}: {len(used)} workgroups, shader clock held {mhz:.0f} MHz')
        assert 500.0 < mhz < 2600.0, mhz
        probe.zero_()
        run_plugin(p, [a.cuda(), w.cuda(), sa.cuda(), sb.cuda()], [out])
        torch.cuda.synchronize()
        assert int(probe.abs().sum().item()) == 0


# ---------------------------------------------------------------------------------------------- decode GEMM on the matrix pipe
class _GemvParams(ctypes.Structure):
    _fields_ = [('wtype', ctypes.c_int32), ('pro', ctypes.c_int32), ('epi', ctypes.c_int32), ('out_dtype', ctypes.c_int32),
                ('M', ctypes.c_int32), ('N', ctypes.c_int32), ('K', ctypes.c_int32), ('x', ctypes.c_void_p), ('ldx', ctypes.c_int64),
                ('w', ctypes.c_void_p), ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('gamma', ctypes.c_void_p), ('eps', ctypes.c_float),
                ('act_scale', ctypes.c_void_p), ('dyn_scale_out', ctypes.c_void_p), ('x_pro_out', ctypes.c_void_p),
                ('residual', ctypes.c_void_p), ('epi_scale', ctypes.c_void_p), ('y', ctypes.c_void_p), ('ldy', ctypes.c_int64)]


@pytest.mark.parametrize('per_channel', [1, 0])
@pytest.mark.parametrize('m,n,k,pro,epi', [
    (8, 12288, 4096, 2, 0),    # RMSNorm + static quantiser -> QKV
    (5, 4096, 4096, 0, 1),     # int8 context -> O + residual
    (8, 11008, 4096, 2, 3),    # RMSNorm + quantiser -> gate | up -> SwiGLU -> static quantiser
    (3, 11008, 4096, 2, 2),    # ... fp16 SwiGLU output
    (8, 4096, 11008, 0, 1),    # int8 intermediate -> down + residual (43 k-steps per wave: a ragged last batch)
    (4, 5120, 5120, 2, 0),     # 13B hidden size: the 6-vector RMSNorm bucket
    (6, 512, 256, 0, 0),       # one k-step per wave, fewer row groups than CUs
    (8, 1024, 13824, 0, 1),    # 13B down-projection rows (124 KB of LDS)
])
def test_mfma_skinny_gemm_equals_the_valu_kernel(m, n, k, pro, epi, per_channel, lib):
    """The SmoothQuant decode GEMM for several sequences on v_mfma_i32_16x16x64_i8 (kernels/gemv_mfma_sq.hip, r04) against the
    skinny vector-ALU kernel (kernels/gemv_impl.h) that it replaces from 5 rows on: exact int32 sums and the same rounding points in
    prologue and epilogue, so every output must be IDENTICAL - for every prologue / epilogue the decode step uses, per-channel and
    per-tensor scales, K blocks that do not divide by the four waves, and M from 3 to 8.  (The VALU kernel itself is held to the oracle by the tests above.)"""
    lib.tllm_gemv.argtypes = [ctypes.POINTER(_GemvParams), ctypes.c_void_p]
    lib.tllm_gemv.restype = ctypes.c_int32
    lib.tllm_gemv_set_mfma_rows.argtypes = [ctypes.c_int32]
    lib.tllm_gemv_set_mfma_rows.restype = None
    g = torch.Generator(device='cuda').manual_seed(m * 1000 + n + k + pro + epi)
    swiglu = epi in (2, 3)
    rows = 2 * n if swiglu else n
    w = torch.randint(-127, 128, (rows, k), dtype=torch.int8, device='cuda', generator=g)
    sc = (torch.rand(rows if per_channel else 1, device='cuda', generator=g) * 2e-3 + 1e-4).float()
    srow = torch.tensor([0.013], device='cuda')
    if pro == 2:
        x = (torch.randn((m, k), device='cuda', generator=g) * 1.7).half()
        gamma = (torch.rand(k, device='cuda', generator=g) + 0.5).half()
        act = torch.tensor([37.0], device='cuda')
    else:
        x = torch.randint(-127, 128, (m, k), dtype=torch.int8, device='cuda', generator=g)
        gamma = act = None
    res = torch.randn((m, n), device='cuda', generator=g).half() if epi == 1 else None
    epi_q = torch.tensor([21.0], device='cuda') if epi == 3 else None
    outs = []
    try:
        for rows_from in (0, 2):  # 0: the vector-ALU kernel; 2: the matrix-pipe kernel for every M >= 2
            lib.tllm_gemv_set_mfma_rows(rows_from)
            y = torch.full((m, n), 7, dtype=torch.int8 if epi == 3 else torch.float16, device='cuda')
            q = _GemvParams(3, pro, epi, 2 if epi == 3 else 1, m, n, k, x.data_ptr(), k, w.data_ptr(), k, sc.data_ptr(), srow.data_ptr(),
                            per_channel, 0, gamma.data_ptr() if gamma is not None else None, 1e-6,
                            act.data_ptr() if act is not None else None, None, None, res.data_ptr() if res is not None else None,
                            epi_q.data_ptr() if epi_q is not None else None, y.data_ptr(), n)
            assert lib.tllm_gemv(ctypes.byref(q), torch.cuda.current_stream().cuda_stream) == 0, capi.last_error()
            torch.cuda.synchronize()
            outs.append(y.cpu().numpy().view(np.int8 if epi == 3 else np.uint16).copy())
    finally:
        lib.tllm_gemv_set_mfma_rows(-1)
    assert np.abs(outs[0].astype(np.int64)).sum() > 0
    np.testing.assert_array_equal(outs[0], outs[1])


# ---------------------------------------------------------------------------------------------- persistent prefill GEMM (r05)
@pytest.mark.parametrize('cfg,m,n,k,residual', [
    (63, 2304, 7000, 640, False),   # 9 x 37 = 333 tiles of 256 x 192 on 256 CUs: two tiles per workgroup, ragged columns, odd K-tile count
    (63, 2100, 7000, 384, True),    # ragged rows as well, the residual in the epilogue (the drained-wait path)
    (62, 2304, 5000, 512, False),   # 9 x 40 = 360 tiles of 256 x 128
    (62, 1300, 12288, 256, True),   # 6 x 96 = 576 tiles: three per workgroup, the shortest K the persistent form serves
    (55, 2304, 7000, 320, True),    # fp16 operands, persistent 256 x 192
    (56, 2304, 5000, 256, False)])  # fp16 operands, persistent 256 x 128
def test_persistent_prefill_gemm_several_tiles_per_workgroup(cfg, m, n, k, residual):
    """The persistent forms of the phased prefill GEMM (gemm_sqp.hip PERSIST: a workgroup walks several tiles, the next tile's first
    K-tiles are requested under the epilogue, the output stores stay counted in the waits) on problems with MORE tiles than CUs -
    test_prefill_gemm_every_tile_shape gives every workgroup one tile.  SmoothQuant: every output exact against an fp64-accumulated
    integer product with the reference's epilogue (fp16(float(acc) * (s_col * s_row)), then the fp16 residual add); fp16: the fp16
    GEMM tolerance.  Ragged last row / column tiles, odd K-tile counts, with and without the fused residual."""
    lib = capi.load_library()

    class GemmParams(ctypes.Structure):
        _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                    ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                    ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                    ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p), ('ldc', ctypes.c_int64)]

    lib.tllm_gemm.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p]
    lib.tllm_gemm.restype = ctypes.c_int32
    lib.tllm_gemm_residual.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p, ctypes.c_void_p]
    lib.tllm_gemm_residual.restype = ctypes.c_int32
    lib.tllm_gemm_set_tile_cfg.argtypes = [ctypes.c_int32]
    lib.tllm_gemm_set_tile_cfg.restype = None
    dev = torch.device('cuda', 0)
    torch.manual_seed(cfg + m)
    sq = cfg >= 60
    stream = torch.cuda.current_stream().cuda_stream
    res = (torch.randn((m, n), device=dev) * 3).half() if residual else None
    c = torch.full((m, n), 7.0, dtype=torch.float16, device=dev)
    if sq:
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
        w = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
        sc = torch.randint(1, 13, (n, ), device=dev).float() * 1e-4
        sr = torch.randint(1, 13, (m, ), device=dev).float() * 1e-3
        q = GemmParams(3, 1, m, n, k, a.data_ptr(), k, w.data_ptr(), k, sc.data_ptr(), sr.data_ptr(), 1, 1, c.data_ptr(), n)
        ref = ((a.double() @ w.double().t()).float() * (sc[None, :] * sr[:, None])).half()
    else:
        a = torch.randn((m, k), dtype=torch.float16, device=dev)
        w = (torch.randn((n, k), device=dev) / np.sqrt(k)).half()
        q = GemmParams(0, 1, m, n, k, a.data_ptr(), k, w.data_ptr(), 2 * k, None, None, 0, 0, c.data_ptr(), n)
        ref = (a.double() @ w.double().t()).half()
    if residual:
        ref = (ref.float() + res.float()).half()
    lib.tllm_gemm_set_tile_cfg(cfg)
    try:
        for _ in range(2):  # twice: the second launch finds a warm instruction cache and different arrival orders
            c.fill_(7.0)
            rc = lib.tllm_gemm_residual(ctypes.byref(q), res.data_ptr(), stream) if residual else lib.tllm_gemm(ctypes.byref(q), stream)
            assert rc == 0, capi.last_error()
            torch.cuda.synchronize()
            if sq:
                assert torch.equal(c, ref), int((c != ref).sum())
            else:
                np.testing.assert_allclose(c.float().cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-3, atol=4e-3 if residual else 2e-3)
    finally:
        lib.tllm_gemm_set_tile_cfg(0)


# ---------------------------------------------------------------------------------------------- split-K-2 prefill GEMM (r06)
@pytest.mark.parametrize('cfg,m,n,k,residual', [
    (64, 1024, 4096, 4096, True),    # the O-projection of a 1024-token prefill: 128 tiles of 256 x 128 -> 256 workgroups, residual epilogue
    (64, 1024, 4096, 11008, True),   # the down-projection: 86 K-tiles, 43 per half
    (64, 1000, 4000, 1664, False),   # ragged rows and columns, 13 K-tiles (7 + 6)
    (64, 300, 1000, 512, False),     # 2 x 8 tiles: a few pairs only
    (65, 1024, 4096, 4096, True),    # 128 x 128 tiles: 512 workgroups, two per CU
    (65, 1000, 4000, 1664, False),
    (65, 512, 4096, 11008, True),    # the down-projection of a 512-token prefill
    (57, 1024, 4096, 4096, True),    # fp16 operands
    (57, 1000, 4000, 832, False),    # fp16, ragged, 13 K-tiles
    (58, 512, 4096, 4096, True),     # fp16, 128 x 128 tiles
    (58, 1000, 4000, 832, False)])
def test_split_k_prefill_gemm_equals_the_one_pass_form(cfg, m, n, k, residual):
    """gemm_sqp.hip KSPLIT (ids 64 / 57): two workgroups per 256 x 128 tile, each half of the K-tiles, the accumulators of the other
    X-half handed to the partner through write-through slabs + a drained flag (guide G16 R1).  SmoothQuant: int32 partial sums are
    exact, so every output equals the fp64-accumulated integer product with the reference's epilogue (cutlass_extensions/.../
    epilogue_per_row_per_col_scale.h:279-347) - and the one-pass kernel's - bit for bit; fp16: the GEMM tolerance.  Launched several
    times in a row (the flags re-arm themselves), on two streams (a workspace per stream)."""
    lib = capi.load_library()

    class GemmParams(ctypes.Structure):
        _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                    ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                    ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                    ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p), ('ldc', ctypes.c_int64)]

    lib.tllm_gemm.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p]
    lib.tllm_gemm.restype = ctypes.c_int32
    lib.tllm_gemm_residual.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p, ctypes.c_void_p]
    lib.tllm_gemm_residual.restype = ctypes.c_int32
    lib.tllm_gemm_set_tile_cfg.argtypes = [ctypes.c_int32]
    lib.tllm_gemm_set_tile_cfg.restype = None
    dev = torch.device('cuda', 0)
    torch.manual_seed(cfg + m + k)
    sq = cfg in (64, 65)
    res = (torch.randn((m, n), device=dev) * 3).half() if residual else None
    if sq:
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
        w = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
        sc = torch.randint(1, 13, (n, ), device=dev).float() * 1e-4
        sr = torch.randint(1, 13, (m, ), device=dev).float() * 1e-3
        acc = torch.zeros((m, n), dtype=torch.float64, device=dev)
        for k0 in range(0, k, 2048):
            acc += a[:, k0:k0 + 2048].double() @ w[:, k0:k0 + 2048].double().t()
        ref = (acc.float() * (sc[None, :] * sr[:, None])).half()
    else:
        a = torch.randn((m, k), dtype=torch.float16, device=dev)
        w = (torch.randn((n, k), device=dev) / np.sqrt(k)).half()
        ref = (a.double() @ w.double().t()).half()
    if residual:
        ref = (ref.float() + res.float()).half()
    streams = [torch.cuda.current_stream(), torch.cuda.Stream(device=dev)]
    outs = []
    for st in streams:
        with torch.cuda.stream(st):
            c = torch.full((m, n), 7.0, dtype=torch.float16, device=dev)
            if sq:
                q = GemmParams(3, 1, m, n, k, a.data_ptr(), k, w.data_ptr(), k, sc.data_ptr(), sr.data_ptr(), 1, 1, c.data_ptr(), n)
            else:
                q = GemmParams(0, 1, m, n, k, a.data_ptr(), k, w.data_ptr(), 2 * k, None, None, 0, 0, c.data_ptr(), n)
            for cfg_now in (cfg, 62 if sq else 56, cfg, cfg):  # split-K, the persistent one-pass form, split-K twice more
                lib.tllm_gemm_set_tile_cfg(cfg_now)
                try:
                    c.fill_(7.0)
                    rc = (lib.tllm_gemm_residual(ctypes.byref(q), res.data_ptr(), st.cuda_stream) if residual
                          else lib.tllm_gemm(ctypes.byref(q), st.cuda_stream))
                    assert rc == 0, capi.last_error()
                    st.synchronize()
                finally:
                    lib.tllm_gemm_set_tile_cfg(0)
                if sq:
                    assert torch.equal(c, ref), (cfg_now, int((c != ref).sum()))
                else:
                    np.testing.assert_allclose(c.float().cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-3, atol=4e-3 if residual else 2e-3)
            outs.append(c.clone())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('cfg', [64, 65])
def test_split_k_gemm_alternating_tile_counts_on_one_workspace(cfg):
    """Regression (r06): the split-K pair flags once sat BEHIND the slabs, so their place moved with the tile count - a launch with
    fewer tiles than an earlier one on the same stream found its flags inside old slab data, skipped the wait, and added a partner's
    sums that were not there yet (seen as a wrong FIRST prefill of a session behind other tests).  Large and small problems alternate
    on one stream, every output exact."""
    lib = capi.load_library()

    class GemmParams(ctypes.Structure):
        _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                    ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                    ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                    ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p), ('ldc', ctypes.c_int64)]

    lib.tllm_gemm.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p]
    lib.tllm_gemm.restype = ctypes.c_int32
    lib.tllm_gemm_set_tile_cfg.argtypes = [ctypes.c_int32]
    lib.tllm_gemm_set_tile_cfg.restype = None
    dev = torch.device('cuda', 0)
    st = torch.cuda.Stream(device=dev)
    probs = []
    for m, n, k in ((1024, 4096, 2048), (200, 4096, 11008), (300, 1000, 512), (1024, 4096, 4096), (64, 4096, 1024)):
        torch.manual_seed(m * 7 + k)
        a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
        w = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
        sc = torch.randint(1, 13, (n, ), device=dev).float() * 1e-4
        sr = torch.randint(1, 13, (m, ), device=dev).float() * 1e-3
        acc = torch.zeros((m, n), dtype=torch.float64, device=dev)
        for k0 in range(0, k, 2048):
            acc += a[:, k0:k0 + 2048].double() @ w[:, k0:k0 + 2048].double().t()
        ref = (acc.float() * (sc[None, :] * sr[:, None])).half()
        c = torch.empty((m, n), dtype=torch.float16, device=dev)
        probs.append((GemmParams(3, 1, m, n, k, a.data_ptr(), k, w.data_ptr(), k, sc.data_ptr(), sr.data_ptr(), 1, 1, c.data_ptr(), n),
                      c, ref, (a, w, sc, sr)))
    torch.cuda.synchronize()
    lib.tllm_gemm_set_tile_cfg(cfg)
    try:
        with torch.cuda.stream(st):
            for rnd in range(6):
                for q, c, ref, _ in probs:
                    c.fill_(3.0)
                    assert lib.tllm_gemm(ctypes.byref(q), st.cuda_stream) == 0, capi.last_error()
                    st.synchronize()
                    assert torch.equal(c, ref), (rnd, q.M, q.N, q.K, int((c != ref).sum()))
    finally:
        lib.tllm_gemm_set_tile_cfg(0)
