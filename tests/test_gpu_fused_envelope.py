"""The one-launch generation path (kernels/qkv_attn_fused.hip: QKV projection + RoPE + cache append + attention, + the
O-projection stage where the session runs it) against the ORACLE across its envelope - not against
the two-launch HIP path (tests/test_gpu_fused_qkv_attn.py does that, bit for bit): one decoder layer at LLaMA-7B dimensions,
batch 1, contexts 3 / 40 of 49 padded / 700 / 2300 / 4000 (fp16 cache: up to 2000), static and per-token SmoothQuant,
weight-only int8 / int4, fp16 weights, int8 and fp16 KV cache.

Both sides start from the SAME cache bytes (the session's synthetic context, read back and handed to the oracle as
`start_caches`), so no 4000-token numpy prefill is needed and what is compared is the generation kernels alone.  Per step:
  * the operand of every GEMM of the layer (`tllm_session_get_tap_ex`) - int8 behind its quantiser for SmoothQuant, in LSBs;
  * the attention context at the reference's generation tolerance atol 2e-3 (T/tests/attention/test_gpt_attention.py:828-831);
  * the bytes the step appends to the KV cache against the oracle's quantiser (T/cpp/tests/runtime/transposeKVKernelTest.cpp:
    98-148 compares element-exact; here +-1 LSB where the two QKV sums straddle a rounding boundary, slot indexing exact);
  * the logits."""
import numpy as np
import pytest

from oracle import quant_oracle as QO
from tensorrt_llm.runtime.native import NativeSession
from test_gpu_bench_geometry import read_cache
from test_gpu_session import synth_model

pytestmark = pytest.mark.gpu

H, D, I, V = 32, 4096, 11008, 512
DH = D // H
STEPS = 4

_models = {}


def model(mode, int8_kv):
    key = (mode, int8_kv)
    if key not in _models:
        cfg, w = synth_model(23, L=1, H=H, D=D, I=I, V=V)
        r = np.random.default_rng(31)
        ids = r.integers(3, V, (1, 64)).astype(np.int32)
        _models[key] = (cfg, QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=np.array([64], np.int32)))
    return _models[key]


# (mode, int8 KV, [(max_input_len, real length)]): the fp16 cache is served up to 2048 slots, the int8 cache up to 4096
CASES = [('sq_static_pc', 1, [(3, 3), (49, 40), (700, 700), (2300, 2300), (4000, 4000)]),
         ('sq_dyn_pc', 1, [(3, 3), (49, 40), (2300, 2300)]),
         ('sq_static_pc', 0, [(49, 40), (700, 700), (2000, 2000)]),
         ('woq8', 1, [(3, 3), (49, 40), (2300, 2300), (4000, 4000)]),
         ('woq8', 0, [(49, 40), (2000, 2000)]),
         # fp16 weights (r06: two 8 KB tiles per row pair), BASELINE.json configs[1] = fp16 + fp16 cache; and with the int8 cache
         ('fp16', 0, [(3, 3), (49, 40), (700, 700), (2000, 2000)]),
         ('fp16', 1, [(49, 40), (2300, 2300)]),
         ('woq4', 1, [(3, 3), (49, 40), (2300, 2300)])]


@pytest.mark.parametrize('mode,int8_kv,shapes', CASES, ids=[f'{m}-kv{"8" if k else "16"}' for m, k, _ in CASES])
def test_one_launch_generation_vs_oracle_across_the_envelope(mode, int8_kv, shapes):
    run_cases(mode, int8_kv, shapes, one_launch=True)


def test_general_launches_at_batch_1_vs_oracle():
    """The same comparison for the launches every OTHER configuration runs (session key fuse_qkv_attention = 0: QKV GEMV, split
    attention + merge, O-projection GEMV), at batch 1 and the 7B dimensions - the leg the one-launch path is held to bit for bit in
    tests/test_gpu_fused_qkv_attn.py, here against the oracle itself."""
    run_cases('sq_static_pc', 1, [(49, 40), (2300, 2300)], one_launch=False)


def run_cases(mode, int8_kv, shapes, one_launch):
    cfg, qmodel = model(mode, int8_kv)
    sq = mode.startswith('sq')
    lw = qmodel['oracle']['layers'][0]
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode'], debug_taps=1, fuse_qkv_attention=-1 if one_launch else 0))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    kv_dtype = np.int8 if int8_kv else np.float16
    for S, length in shapes:
        NEW = STEPS + 1
        smax = S + NEW
        s.setup(1, S, NEW)
        if one_launch:
            assert s.decode_form() & 1, 'this geometry must take the one-launch projection + attention'
            if mode == 'sq_static_pc' or mode == 'woq8':
                assert s.decode_form() & 2, 'static SmoothQuant / weight-only int8: the O-projection stage must be on'
        else:
            assert s.decode_form() & 3 == 0
        s.fake_context(length, seed=5 + S)
        start = read_cache(s, 0, (1, 2, H, smax, DH), kv_dtype)
        if not int8_kv:
            assert np.isfinite(start.astype(np.float32)).all()
        # the synthetic cache holds uniform random values up to vmax (int8: the full +-127 range x the dequantisation scale, several
        # times what a real V row holds): the reference's absolute tolerances are stated for O(1) data and scale with it - the
        # probabilities are rounded to fp16 relative to the lane group's own maximum here and to the row's in the oracle, a
        # relative 2^-11 per term that does not average out over a 3-token context
        vmax = max(1.0, float(np.abs(start[:, 1, :, :S].astype(np.float32)).max()) * (float(lw['kv_qo']) if int8_kv else 1.0))
        got_logits, got_taps = [], []
        for i in range(STEPS):
            s.step(1, use_graph=(i >= 2))
            got_logits.append(s.logits())
            got_taps.append({n: s.tap(0, n, {'qkv_in': D, 'o_in': D, 'mlp_in': D, 'proj_in': I}[n], quantised=sq)
                             for n in ('qkv_in', 'o_in', 'mlp_in', 'proj_in')})
        out = s.output_ids()
        end = read_cache(s, 0, (1, 2, H, smax, DH), kv_dtype)
        # ---- oracle from the same cache bytes, fed the session's own tokens (step 0 consumes the synthetic context's token 3)
        feed = np.concatenate([np.full((1, 1), 3, np.int32), out[:, S + 1:S + STEPS]], axis=1)
        ids = np.full((1, S), 3, np.int32)
        lens = np.array([length], np.int32)
        taps = {}
        ref, _ = QO.run_model(qmodel, ids, lens, NEW, feed_ids=feed, taps=taps, start_caches=[start])
        scale = max(max(np.abs(r).max() for r in ref[1:]), 1.0)
        for i in range(STEPS):
            tag = f'[{mode} kv{"8" if int8_kv else "16"} S={S} len={length}] step {i}'
            oin = taps['gemm_in'][i][0]
            for n in ('qkv_in', 'o_in', 'mlp_in', 'proj_in'):
                g, w_ = got_taps[i][n][0], oin[n][0]
                if sq:
                    d = np.abs(g.astype(np.int32) - w_.astype(np.int32))
                    same = float(np.mean(d == 0))
                    # qkv_in depends on nothing the kernels computed: identical.  The attention context: +-1 LSB where the fp32 sums of
                    # the two implementations straddle a quantiser boundary.  Behind the O-projection every element of x + O(ctx)
                    # may sit one fp16 ulp apart (a one-LSB context element moves all 4096 sums), and one ulp of a value near 4 is
                    # 5 - 10 % of post_layernorm's quantiser step: more flips, never more than one LSB
                    # (proj_in: one fp16 ulp of silu(fc) * gate near its largest values spans two steps of the SwiGLU quantiser)
                    floor = dict(qkv_in=1.0, o_in=0.95, mlp_in=0.85, proj_in=0.85)[n]
                    assert d.max() <= dict(qkv_in=0, o_in=1, mlp_in=1, proj_in=2)[n] and same >= floor, (tag, n, int(d.max()), same)
                else:
                    tol = dict(qkv_in=(1e-3, 1e-3), o_in=(2e-3 * vmax, 1e-3), mlp_in=(8e-3 * vmax, 4e-3), proj_in=(8e-3 * vmax, 8e-3))[n]
                    np.testing.assert_allclose(g.astype(np.float32), w_.astype(np.float32), atol=tol[0], rtol=tol[1], err_msg=f'{tag} {n}')
            # the attention context before the O-projection's quantiser
            octx = taps['attn_ctx'][i][0][0]
            if sq:
                if 'dyn' in mode:
                    oq = QO.O.quantize_per_token(octx[None])[0][0].astype(np.int32)
                else:
                    oq = QO.O.quantize_tensor(octx, lw['attn_qscale']).astype(np.int32)
                assert np.abs(got_taps[i]['o_in'][0].astype(np.int32) - oq).max() <= 1, tag
            dl = np.abs(got_logits[i] - ref[i + 1])
            assert np.isfinite(got_logits[i]).all(), tag
            np.testing.assert_allclose(got_logits[i], ref[i + 1], atol=(5e-2 if sq else 5e-3) * scale, err_msg=tag)
            assert dl.mean() < (1.2e-2 if sq else 1e-3) * scale, (tag, float(dl.mean()))
        # ---- the cache: nothing but slots S .. S + STEPS - 1 changed, and those hold what the oracle's quantiser stored
        ocache = taps['caches'][0]
        keep = np.ones(smax, bool)
        keep[S:S + STEPS] = False
        np.testing.assert_array_equal(end[:, :, :, keep], start[:, :, :, keep])
        got = end[:, :, :, S:S + STEPS].astype(np.float32)
        want = ocache[:, :, :, S:S + STEPS].astype(np.float32)
        if int8_kv:
            d = np.abs(got - want)
            assert d.max() <= 1 and np.mean(d == 0) > 0.97, (mode, S, float(d.max()), float(np.mean(d == 0)))
        else:
            np.testing.assert_allclose(got, want, atol=4e-3, rtol=4e-3)
        print(f'[{mode} kv{"8" if int8_kv else "16"} S={S} len={length}] {STEPS} steps: operands, context, logits (max |d| '
              f'{max(np.abs(got_logits[i] - ref[i + 1]).max() for i in range(STEPS)):.3g} of scale {scale:.3g}), appended cache rows OK')
    s.close()
