"""The gated MLP of the batch-1 decode step in ONE launch (kernels/mlp_fused.hip, r06: RMSNorm + gate|up + SwiGLU + quantiser +
down-projection + residual, the intermediate row handed over inside the launch) against the two GEMV launches it replaces
(the default; the one-launch form is opt-in, session key fuse_mlp = 1: gemv_kernel<W_INT8_SQ, PK_NORM, EK_SWIGLU> + gemv_ksplit_kernel<W_INT8_SQ>), at the LLaMA-7B layer
dimensions - the geometry the launch is built for.

Reference semantics of what is fused: GatedMLP.forward (PY/layers/mlp.py:43-73) behind RmsNorm (PY/layers/normalization.py:33-54)
with the static SmoothQuant quantisers (K/quantization.cu:31-59) and the SmoothQuant GEMM epilogue
(cutlass_extensions/.../epilogue_per_row_per_col_scale.h:279-347).

The launch restates both kernels value for value - the prologue's summation order, exact integer dot products, the same epilogue
expressions - so EVERYTHING must be identical: the int8 operand behind post_layernorm (tap mlp_in), the quantised SwiGLU row (tap
proj_in), every logit, every token, every byte of the KV cache.  Eager steps and graph replays; many launches in a row on one
exchange area (the tag of a launch is the previous launch's + 1, kept in the area itself); the bounded wait's error path."""
import numpy as np
import pytest

from tensorrt_llm.runtime.native import NativeSession
from test_gpu_fused_qkv_attn import make, read_cache, weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('S,pad,int8_kv,front', [(3, 0, 1, 1), (40, 9, 1, 1), (1100, 0, 1, 1), (300, 5, 0, 1), (700, 0, 1, 0)])
def test_one_launch_mlp_equals_the_two_gemv_launches(S, pad, int8_kv, front):
    layers, NEW = 2, 7
    cfg, w, qm = weights(layers, int8_kv)
    D, I = cfg['hidden_size'], cfg['inter_size']
    max_in = S + pad
    r = np.random.default_rng(500 + S)
    ids = np.full((1, max_in), 2, np.int32)
    ids[0, :S] = r.integers(3, cfg['vocab_size'], S)
    lens = np.array([S], np.int32)
    out = {}
    for mlp in (0, 1):
        s = make(cfg, w, qm, front, fuse_mlp=mlp)
        s.setup(1, max_in, NEW)
        assert bool(s.decode_form() & 4) == bool(mlp), s.decode_form()
        s.context(ids, lens)
        rec = dict(mlp_in=[], proj_in=[], logits=[s.logits()])
        for i in range(NEW - 1):
            s.step(1, use_graph=i >= 2)
            rec['mlp_in'].append(np.stack([s.tap(li, 'mlp_in', D, quantised=True)[0] for li in range(layers)]))
            rec['proj_in'].append(np.stack([s.tap(li, 'proj_in', I, quantised=True)[0] for li in range(layers)]))
            rec['logits'].append(s.logits())
        rec['tokens'] = s.output_ids()
        nbytes = 2 * cfg['num_heads'] * (max_in + NEW) * (D // cfg['num_heads']) * (1 if int8_kv else 2)
        rec['cache'] = [read_cache(s, li, nbytes) for li in range(layers)]
        out[mlp] = rec
        s.close()
    a, b = out[0], out[1]
    for i in range(NEW - 1):
        np.testing.assert_array_equal(a['mlp_in'][i], b['mlp_in'][i], err_msg=f'step {i}: the operand behind post_layernorm')
        np.testing.assert_array_equal(a['proj_in'][i], b['proj_in'][i], err_msg=f'step {i}: the quantised SwiGLU row')
    for i in range(NEW):
        np.testing.assert_array_equal(a['logits'][i], b['logits'][i], err_msg=f'logits {i}')
    np.testing.assert_array_equal(a['tokens'], b['tokens'])
    for li in range(layers):
        np.testing.assert_array_equal(a['cache'][li], b['cache'][li])


def test_one_launch_mlp_many_launches_graph_and_eager():
    """32 layers x 40 steps = 1280 launches on one exchange area: graph replay, eager launches and the two-GEMV form give the same
    tokens and logits; a second request on the same session starts from whatever tag the first left."""
    cfg, w, qm = weights(32, 1)
    S, NEW = 600, 40
    lens = np.array([S], np.int32)
    ids = np.random.default_rng(19).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    ref = make(cfg, w, qm, 1, taps=False, fuse_mlp=0)
    ref.setup(1, S, NEW)
    want = ref.generate(ids, lens, NEW)
    want_logits = ref.logits()
    s = make(cfg, w, qm, 1, taps=False, fuse_mlp=1)
    s.setup(1, S, NEW)
    assert s.decode_form() == 7
    np.testing.assert_array_equal(s.generate(ids, lens, NEW), want)
    np.testing.assert_array_equal(s.logits(), want_logits)
    s.setup(1, S, NEW)
    s.context(ids, lens)
    s.step(NEW - 1, use_graph=False)
    np.testing.assert_array_equal(s.output_ids(), want)
    np.testing.assert_array_equal(s.logits(), want_logits)
    ids2 = np.random.default_rng(20).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    ref.setup(1, S, NEW)
    want2 = ref.generate(ids2, lens, NEW)
    s.setup(1, S, NEW)
    np.testing.assert_array_equal(s.generate(ids2, lens, NEW), want2)
    assert s.fused_retries() == 0 and s.decode_form() == 7
    ref.close()
    s.close()


def test_expired_wait_of_the_one_launch_mlp_falls_back_and_repeats_the_request():
    """`fused_max_spins = 0`: the first look at the members' / groups' lines cannot find them all written - the error word is raised (bit 16), later
    launches return at entry, the session drops the one-launch forms at its next synchronisation and tllm_session_generate runs the
    request AGAIN on the GEMV launches.  The caller sees the tokens of a session that never used the one-launch MLP."""
    cfg, w, qm = weights(4, 1)
    S, NEW = 200, 40
    ids = np.random.default_rng(78).integers(3, cfg['vocab_size'], (1, S)).astype(np.int32)
    lens = np.array([S], np.int32)
    ref = make(cfg, w, qm, 0, taps=False, fuse_mlp=0)
    ref.setup(1, S, NEW)
    want = ref.generate(ids, lens, NEW)
    ref.close()
    s = make(cfg, w, qm, 0, taps=False, fuse_mlp=1, fused_max_spins=0)  # (two-launch attention: only the MLP's wait can expire)
    s.setup(1, S, NEW)
    assert s.decode_form() == 4
    got = s.generate(ids, lens, NEW)
    assert s.fused_retries() == 1 and s.decode_form() == 0
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(s.generate(ids, lens, NEW), want)
    assert s.fused_retries() == 1
    # a new setup switches the one-launch form on again (its flags are cleared) and it runs through with the default bound
    s.close()
