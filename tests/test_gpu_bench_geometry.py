"""Oracle parity at the geometry bench.py times (BASELINE.json configs[1..3] at batch 1, context ~1024 + the tokens generated
so far): one decoder layer at LLaMA-7B dimensions (D = 4096, 32 heads of 128, I = 11008) behind a real 1150-token prefill,
then four generation steps - two eager, two replayed from the step's hipGraph.

What runs that the small-shape tests do not reach together: for SmoothQuant at the 7B dimensions the one-launch QKV projection
+ RoPE + cache append + attention (`qkv_attn_fused_kernel<3, int8>`: 8 workgroups per head, a cache capacity of 1156 slots, 6 of
the 8 members holding used slots; the 13B rows take the two launches); for the other modes `mmha_partial_kernel<128, 12, int8 |
fp16>` with 7 split slots (6 active at lengths 1150 / 1151, all 7 at 1152 / 1153) and the merge by the last split of a head to
arrive inside that launch; the static SmoothQuant quantiser behind either, the K-split down-projection at K = 11008, and the
prefill GEMMs at M = 1150 (256 x 192 / 128 x 128 MFMA tiles with a ragged last row tile).

Checked per step: the O-projection's INPUT (the merged attention context, `tllm_session_get_tap`) at the reference's
generation-attention tolerance atol 2e-3 (T/tests/attention/test_gpt_attention.py:828-831) - for SmoothQuant the tap is the
int8 tensor behind the static quantiser, compared in LSBs - and the logits.  The generation steps start from the ORACLE's KV
cache (copied over the session's after comparing the two), so what is compared is the generation kernels, not the
accumulated +-1 LSB differences of two 1150-token prefills."""
import ctypes

import numpy as np
import pytest

from oracle import quant_oracle as QO
from tensorrt_llm.runtime.native import NativeSession
from test_gpu_session import synth_model

pytestmark = pytest.mark.gpu

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL('libamdhip64.so')
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    return _hip


def read_cache(s, layer, shape, dtype):
    host = np.empty(shape, dtype)
    assert hip().hipMemcpy(host.ctypes.data, s.kv_cache_ptr(layer), host.nbytes, 2) == 0  # device -> host
    return host


def write_cache(s, layer, host):
    host = np.ascontiguousarray(host)
    assert hip().hipMemcpy(s.kv_cache_ptr(layer), host.ctypes.data, host.nbytes, 1) == 0  # host -> device


# LLaMA-13B layer dimensions (D = 5120, 40 heads, I = 13824; T/examples/llama/README: --n_embd 5120 --n_head 40 --inter_size 13824):
# rows of 5 / 10 / 13.5 KiB - the GEMV takes its second activation-vector bucket, the down-projection is beyond the K-split
# kernel's 12 KiB and runs the general one, 40 heads x 3 splits do not fill the chip - behind a 330-token prefill
@pytest.mark.parametrize('mode,int8_kv,dims', [('sq_static_pc', 1, '7b'), ('woq8', 1, '7b'), ('fp16', 0, '7b'),
                                               ('sq_static_pc', 1, '13b'), ('woq8', 1, '13b'), ('fp16', 0, '13b'),
                                               ('sq_dyn_pc', 1, '13b')])
def test_decoder_layer_at_the_bench_geometry_vs_oracle(mode, int8_kv, dims):
    H, D, I = (32, 4096, 11008) if dims == '7b' else (40, 5120, 13824)
    cfg, w = synth_model(23, L=1, H=H, D=D, I=I, V=512)
    Dh = D // H
    B, S, NEW = 1, (1150 if dims == '7b' else 330), 6
    STEPS = 4
    smax = S + NEW
    r = np.random.default_rng(31)
    ids = r.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    lens = np.array([S], np.int32)
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids[:, :96], calib_lens=np.array([96], np.int32))
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode'], debug_taps=1))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    logits_ctx = s.logits()
    first = s.output_ids()[:, S]
    # ---- oracle: prefill + STEPS generation steps fed with the session's own greedy ids (filled in below, step by step:
    # the ids of step i only enter the oracle's step i + 1, so one oracle pass after the GPU run is enough)
    kv_dtype = np.int8 if int8_kv else np.float16
    gpu_cache_ctx = read_cache(s, 0, (B, 2, H, smax, Dh), kv_dtype)

    # a first oracle pass for the context phase only: its cache seeds the session's generation steps
    taps0 = {}
    ref0, _ = QO.run_model(qmodel, ids, lens, 1, taps=taps0)
    scale = max(np.abs(ref0[0]).max(), 1.0)
    sq = mode.startswith('sq')
    d_ctx = np.abs(logits_ctx - ref0[0])
    print(f'\n[{mode}] context logits: max |d| = {d_ctx.max():.4g}, mean |d| = {d_ctx.mean():.4g}, scale = {scale:.4g}')
    # observed on MI355X: SmoothQuant max 3.4e-2 / mean 8.9e-3 of the logit range (+-1 LSB quantiser flips amplified by
    # the following GEMMs), fp16 / weight-only max 9e-4 / mean 2.5e-4
    np.testing.assert_allclose(logits_ctx, ref0[0], atol=(5e-2 if sq else 5e-3) * scale)
    assert d_ctx.mean() < (1.2e-2 if sq else 1e-3) * scale
    ocache = taps0['caches_after_context'][0]  # [B, 2, H, S + 1, Dh]
    got = gpu_cache_ctx[:, :, :, :S].astype(np.float32)
    want = ocache[:, :, :, :S].astype(np.float32)
    if int8_kv:
        # indexing is bit-exact (every slot holds ITS token's value); the values themselves are +-1 LSB where the fp16 RoPE /
        # GEMM rounding of the two implementations straddles a quantisation boundary (fma contraction, summation order)
        d = np.abs(got - want)
        print(f'[{mode}] int8 KV after the prefill: {np.mean(d == 0) * 100:.2f} % identical, max {d.max():.0f} LSB')
        assert d.max() <= 1 and np.mean(d == 0) > 0.97
    else:
        np.testing.assert_allclose(got, want, atol=4e-3, rtol=4e-3)
    assert not gpu_cache_ctx[:, :, :, S:].any(), 'slots beyond the prompt must be untouched'
    # ---- generation from the oracle's cache
    seed = np.zeros((B, 2, H, smax, Dh), kv_dtype)
    seed[:, :, :, :S] = ocache[:, :, :, :S]
    write_cache(s, 0, seed)
    got_logits, got_taps = [], []
    for i in range(STEPS):
        s.step(1, use_graph=(i >= 2))
        got_logits.append(s.logits())
        got_taps.append(s.attention_tap(0, D, quantised=sq))
    out = s.output_ids()
    s.close()
    np.testing.assert_array_equal(out[:, S], first)
    taps = {}
    ref, _ = QO.run_model(qmodel, ids, lens, STEPS + 1, feed_ids=out[:, S:S + STEPS + 1], taps=taps)
    lw = qmodel['oracle']['layers'][0]
    for i in range(STEPS):
        octx = taps['attn_ctx'][i][0]  # [B, H * Dh] fp16 values
        if sq:
            if 'dyn' in mode:  # per-token activation scales: amax / 127 of the row (K/quantization.cu:94-118)
                oq, osc = QO.O.quantize_per_token(octx)
                oq, lsb = oq.astype(np.int32), float(np.max(osc))
            else:
                oq = QO.O.quantize_tensor(octx, lw['attn_qscale']).astype(np.int32)
                lsb = 1.0 / float(lw['attn_qscale'])
            d = np.abs(got_taps[i].astype(np.int32) - oq)
            print(f'[{mode}] step {i} (length {S + i}): O-projection input int8 {np.mean(d == 0) * 100:.2f} % identical, max {d.max()} '
                  f'LSB (1 LSB = {lsb:.3g})')
            assert d.max() <= 1 and np.mean(d == 0) > 0.97
        else:
            d = np.abs(got_taps[i].astype(np.float32) - octx)
            print(f'[{mode}] step {i} (length {S + i}): O-projection input max |d| = {d.max():.3g}')
            # the reference's bound (atol 2e-3 on O(1) data) + one fp16 ulp where |ctx| > 2 (ulp(2..4) = 1.95e-3)
            np.testing.assert_allclose(got_taps[i].astype(np.float32), octx, atol=2e-3, rtol=1e-3)
        dl = np.abs(got_logits[i] - ref[i + 1])
        print(f'[{mode}] step {i}: logits max |d| = {dl.max():.4g}, mean |d| = {dl.mean():.4g} (scale {scale:.4g})')
        assert np.isfinite(got_logits[i]).all()
        np.testing.assert_allclose(got_logits[i], ref[i + 1], atol=(5e-2 if sq else 5e-3) * scale)
        assert dl.mean() < (1.2e-2 if sq else 1e-3) * scale


def test_distance_to_the_reference_rounding_points():
    """The HIP kernels round LATER than the reference at two places (fewer roundings): the weight-only
    GEMV applies the fp16 scale once to the fp32 sum (reference: fp16(q * s), then fp16(x * w) per product,
    K/weightOnlyMatrixVectorMultiplication.cu:42-53,187) and the split-KV attention keeps fp32 partial outputs (reference:
    16 partial sums staged through fp16, MM/...Template.h:1957-1980).  The oracle restates both; this reports the three
    distances (HIP - oracle as built, HIP - reference rounding, between the two oracles) on one 7B-dimension layer and holds
    the HIP path to the reference's own tolerances against BOTH."""
    cfg, w = synth_model(23, L=1, H=32, D=4096, I=11008, V=512)
    D = 4096
    B, S, NEW = 2, 40, 3
    r = np.random.default_rng(3)
    ids = r.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    lens = np.array([S, 29], np.int32)
    for b in range(B):
        ids[b, lens[b]:] = 2
    qmodel = QO.quantise_model(cfg, w, 'woq8', 1, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode'], debug_taps=1))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    got_l, got_t = [], []
    for i in range(2):
        s.step(1, use_graph=False)
        got_l.append(s.logits())
        got_t.append(s.attention_tap(0, D, quantised=False).astype(np.float32))
    out = s.output_ids()
    s.close()
    res = {}
    for name, rr in (('as_built', False), ('reference_rounding', True)):
        taps = {}
        ref, _ = QO.run_model(qmodel, ids, lens, 3, feed_ids=out[:, S:S + 3], taps=taps, reference_rounding=rr)
        res[name] = (ref, taps)
    scale = max(np.abs(res['as_built'][0][0]).max(), 1.0)
    for i in range(2):
        t_a, t_r = res['as_built'][1]['attn_ctx'][i][0], res['reference_rounding'][1]['attn_ctx'][i][0]
        l_a, l_r = res['as_built'][0][i + 1], res['reference_rounding'][0][i + 1]
        print(f'\nstep {i}: attention out  |hip - as_built| {np.abs(got_t[i] - t_a).max():.3g}  |hip - ref_rounding| '
              f'{np.abs(got_t[i] - t_r).max():.3g}  |as_built - ref_rounding| {np.abs(t_a - t_r).max():.3g}')
        print(f'step {i}: logits         |hip - as_built| {np.abs(got_l[i] - l_a).max():.3g}  |hip - ref_rounding| '
              f'{np.abs(got_l[i] - l_r).max():.3g}  |as_built - ref_rounding| {np.abs(l_a - l_r).max():.3g}  (scale {scale:.3g})')
        # the reference's generation-attention tolerance (atol 2e-3 on O(1) data, + one fp16 ulp relative: these outputs reach
        # |4|) against the oracle the kernels follow.  The reference's own rounding points put ITS result up to ~4e-3 away
        # from the single-rounding value here (16 partial sums rounded to fp16 before they are added), so against that the
        # claim is: the HIP result is no further from the reference's rounding than exact arithmetic is, + the same tolerance
        # ... for all but a handful of elements: a V (or K) row whose fp16 value sits on an int8 rounding tie is stored one
        # LSB apart by the HIP path and the oracle (their QKV sums differ in the last fp16 ulp), and one LSB of V (~4/127)
        # times that token's attention weight reaches the output - seen: 1 element of 8192 at 2.4e-3.  Those are held to
        # one such step (4e-3), and there may only be a few of them.
        d_a = np.abs(got_t[i] - t_a)
        over = d_a > 2e-3 + 1e-3 * np.abs(t_a)
        assert over.sum() <= 8 and d_a.max() <= 4e-3, (int(over.sum()), float(d_a.max()))
        between = np.abs(t_a - t_r).max()
        assert np.abs(got_t[i] - t_r).max() <= between + 4e-3, (np.abs(got_t[i] - t_r).max(), between)
        np.testing.assert_allclose(got_l[i], l_a, atol=3e-2 * scale)
        assert np.abs(got_l[i] - l_r).max() <= np.abs(l_a - l_r).max() + 1e-2 * scale
