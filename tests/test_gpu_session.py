"""GPU parity of the C++ host loop (include/tllm_runtime_api.h): the whole LLaMA path — embedding, fused
generation step (RMSNorm/quant/SwiGLU/residual folded into the GEMVs), attention plugin kernels, head, greedy
sampler — against (a) the HF-CPU golden logits of tests/golden/hf_tiny_llama.npz with the reference's own bound
(atol 1e-1, T/tests/model/test_llama.py:286-288,352-354) and (b) the numpy oracle."""
import os

import functools

import numpy as np
import pytest

from oracle import llama_oracle as O
from oracle import quant_oracle as QO
from tensorrt_llm.runtime.native import NativeSession

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_tiny():
    t = dict(np.load(os.path.join(GOLD, 'hf_tiny_llama.npz')))
    w = {k: t[k] for k in ('vocab_embedding.weight', 'ln_f.weight', 'lm_head.weight')}
    for k in t:
        if k.startswith('layers.'):
            w[k] = t[k]
    return t, w


TINY_CFG = dict(num_layers=2, num_heads=2, hidden_size=64, inter_size=24, vocab_size=128, max_position_embeddings=64,
                rms_norm_eps=1e-6)


def oracle_weights(w):
    ow = {k: w[k].astype(np.float32) for k in ('vocab_embedding.weight', 'ln_f.weight', 'lm_head.weight')}
    ow['layers'] = []
    for i in range(TINY_CFG['num_layers']):
        pre = f'layers.{i}.'
        ow['layers'].append({k[len(pre):]: w[k].astype(np.float32) for k in w if k.startswith(pre)})
    return ow


def test_tiny_llama_fp16_logits_vs_hf_and_oracle():
    t, w = load_tiny()
    s = NativeSession(dict(TINY_CFG, quant_mode=0))
    for k, v in w.items():
        s.set_tensor(k, v)
    s.finalize()
    ids, lens = t['ids'], t['input_lengths']
    B, S = ids.shape
    s.setup(B, S, 4)
    s.context(ids, lens)
    logits = s.logits()
    np.testing.assert_allclose(logits, t['logits_ctx'], atol=1e-1)  # reference bound
    assert np.abs(logits - t['logits_ctx']).max() < 3e-2            # what fp16 storage actually gives
    out = s.output_ids()
    np.testing.assert_array_equal(out[:, S], t['next_ids'])
    np.testing.assert_array_equal(out[:, :S], ids)
    # oracle: same fp16 rounding points -> much tighter
    H, Dh, smax = 2, 32, S + 4
    ow = oracle_weights(w)
    caches = [np.zeros((B, 2, H, smax, Dh), np.float16) for _ in range(2)]
    ol = O.llama_logits_context(ids, ow, caches, lens, H)
    np.testing.assert_allclose(logits, ol, atol=2e-2)
    # one generation step, eager
    s.step(1, use_graph=False)
    l2 = s.logits()
    np.testing.assert_allclose(l2, t['logits_dec'], atol=1e-1)
    assert np.abs(l2 - t['logits_dec']).max() < 3e-2
    masked = np.zeros((B, smax), np.int32)
    for b in range(B):
        masked[b, lens[b]:S] = 1
    ol2 = O.llama_logits_decode(t['next_ids'], ow, caches, [S, S], lens, S, S, H, masked)
    np.testing.assert_allclose(l2, ol2, atol=2e-2)
    # KV cache contents after context + 1 step vs the oracle's (atol/rtol of test_gpt_attention.py:561-578)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    for li in range(2):
        n = B * 2 * H * smax * Dh
        host = np.empty(n, np.float16)
        assert hip.hipMemcpy(host.ctypes.data, s.kv_cache_ptr(li), n * 2, 2) == 0  # hipMemcpyDeviceToHost
        got = host.reshape(B, 2, H, smax, Dh).astype(np.float32)
        ref = caches[li].astype(np.float32)
        for b in range(B):
            valid = list(range(int(lens[b]))) + [S]
            np.testing.assert_allclose(got[b][:, :, valid], ref[b][:, :, valid], atol=2e-3, rtol=2e-3)
    s.close()


def test_force_tokens_feeds_the_next_step():
    """tllm_session_force_tokens (teacher forcing for parity tests): the forced ids replace the sampler's choice in the output
    buffer, as the next step's input id and as its input embedding row - the following step's logits are the oracle's for the
    forced token, eager and replayed from the step graph, and the step counters are untouched."""
    t, w = load_tiny()
    ids, lens = t['ids'], t['input_lengths']
    B, S = ids.shape
    H, Dh, smax = 2, 32, S + 6
    ow = oracle_weights(w)
    masked = np.zeros((B, smax), np.int32)
    for b in range(B):
        masked[b, lens[b]:S] = 1
    s = NativeSession(dict(TINY_CFG, quant_mode=0))
    for k, v in w.items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, 6)
    s.context(ids, lens)
    caches = [np.zeros((B, 2, H, smax, Dh), np.float16) for _ in range(2)]
    O.llama_logits_context(ids, ow, caches, lens, H)
    r = np.random.default_rng(5)
    seq0 = s.step_state()['sequence_length'].copy()
    for step in range(4):
        forced = r.integers(3, TINY_CFG['vocab_size'], B).astype(np.int32)
        assert not np.array_equal(forced, s.output_ids()[:, S + step])  # a real override
        s.force_tokens(forced)
        np.testing.assert_array_equal(s.output_ids()[:, S + step], forced)
        np.testing.assert_array_equal(s.step_state()['sequence_length'], seq0 + step)  # counters untouched by the override
        s.step(1, use_graph=step >= 2)
        ref = O.llama_logits_decode(forced, ow, caches, [S + step] * B, lens, S, S + step, H, masked)
        np.testing.assert_allclose(s.logits(), ref, atol=2e-2)
    s.close()


def test_generate_graph_equals_eager():
    """The captured generation step must reproduce the eager one token for token."""
    t, w = load_tiny()
    ids, lens = t['ids'], t['input_lengths']
    B, S = ids.shape
    outs = []
    for use_graph in (False, True):
        s = NativeSession(dict(TINY_CFG, quant_mode=0))
        for k, v in w.items():
            s.set_tensor(k, v)
        s.finalize()
        s.setup(B, S, 24)
        if use_graph:
            outs.append(s.generate(ids, lens, 24, end_id=-1))
        else:
            s.context(ids, lens)
            s.step(23, use_graph=False)
            outs.append(s.output_ids())
        s.close()
    np.testing.assert_array_equal(outs[0], outs[1])
    # greedy continuation must also match the oracle's greedy loop for the first steps
    H, Dh, smax = 2, 32, S + 24
    ow = oracle_weights(w)
    caches = [np.zeros((B, 2, H, smax, Dh), np.float16) for _ in range(2)]
    cur = O.llama_logits_context(ids, ow, caches, lens, H).argmax(-1)
    masked = np.zeros((B, smax), np.int32)
    for b in range(B):
        masked[b, lens[b]:S] = 1
    want = [cur]
    for step in range(5):
        lg = O.llama_logits_decode(cur, ow, caches, [S + step] * B, lens, S, S + step, H, masked)
        cur = lg.argmax(-1)
        want.append(cur)
    np.testing.assert_array_equal(outs[1][:, S:S + 6], np.stack(want, 1))


def test_tp_collectives_inside_the_captured_graph_single_rank():
    """The tensor-parallel exchange steps (64 all-reduces + 1 all-gather per 7B step) are RCCL calls enqueued on the
    session's stream and captured into the step's hipGraph.  A 1-GPU box cannot run 2 ranks, but it can run the very
    same call sequence on a 1-rank communicator (`force_comm`): eager == graph == the run without collectives."""
    import ctypes
    from tensorrt_llm.plugin import capi
    lib = capi.load_library()
    uid = (ctypes.c_char * 128)()
    assert lib.tllm_comm_get_unique_id(uid) == 0, capi.last_error()
    assert lib.tllm_comm_init_rank((ctypes.c_int32 * 1)(0), 1, 0, uid) == 0, capi.last_error()
    try:
        t, w = load_tiny()
        ids, lens = t['ids'], t['input_lengths']
        B, S = ids.shape
        outs = {}
        for name, cfg, graph in (('plain', {}, True), ('comm_eager', {'force_comm': 1}, False), ('comm_graph', {'force_comm': 1}, True)):
            s = NativeSession(dict(TINY_CFG, quant_mode=0, **cfg))
            for k, v in w.items():
                s.set_tensor(k, v)
            s.finalize()
            s.setup(B, S, 40)
            if graph:
                outs[name] = s.generate(ids, lens, 40, end_id=-1)  # context, 1 eager step, then 32-step graph chunks
            else:
                s.context(ids, lens)
                s.step(39, use_graph=False)
                outs[name] = s.output_ids()
            s.close()
        np.testing.assert_array_equal(outs['comm_eager'], outs['plain'])
        np.testing.assert_array_equal(outs['comm_graph'], outs['plain'])
    finally:
        assert lib.tllm_comm_destroy_all() == 0


@functools.lru_cache(maxsize=4)
def synth_model(seed, L=2, H=4, D=256, I=512, V=512):
    """(cached: a 7B- / 65B-dimension layer is drawn once per process, not once per parametrisation - callers do not modify it)"""
    r = np.random.default_rng(seed)
    xav = lambda n, k: r.uniform(-1, 1, (n, k)) * np.sqrt(6.0 / (n + k)) * 2
    w = {'vocab_embedding.weight': r.standard_normal((V, D)) * 0.5, 'ln_f.weight': 1 + 0.1 * r.uniform(-1, 1, D),
         'lm_head.weight': xav(V, D)}
    for i in range(L):
        p = f'layers.{i}.'
        w[p + 'input_layernorm.weight'] = 1 + 0.1 * r.uniform(-1, 1, D)
        w[p + 'post_layernorm.weight'] = 1 + 0.1 * r.uniform(-1, 1, D)
        w[p + 'attention.qkv.weight'] = xav(3 * D, D)
        w[p + 'attention.dense.weight'] = xav(D, D)
        w[p + 'mlp.fc.weight'] = xav(I, D)
        w[p + 'mlp.gate.weight'] = xav(I, D)
        w[p + 'mlp.proj.weight'] = xav(D, I)
    w = {k: v.astype(np.float16) for k, v in w.items()}
    cfg = dict(num_layers=L, num_heads=H, hidden_size=D, inter_size=I, vocab_size=V, max_position_embeddings=128,
               rms_norm_eps=1e-6)
    return cfg, w


@pytest.mark.parametrize('mode', ['woq8', 'woq4', 'sq_static', 'sq_static_pc', 'sq_dyn', 'sq_dyn_pc'])
@pytest.mark.parametrize('int8_kv', [0, 1])
def test_quantised_paths_vs_oracle(mode, int8_kv):
    """Weight-only int8/int4 and SmoothQuant (static / per-token x per-tensor / per-channel), with fp16 and int8
    KV cache: context logits and 3 generation steps against the quantisation oracle (oracle/quant_oracle.py)
    on the same quantised weights and scales."""
    cfg, w = synth_model(11)
    B, S, NEW = 2, 12, 4
    r = np.random.default_rng(5)
    ids = r.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    lens = np.array([S, 7], np.int32)
    for b in range(B):
        ids[b, lens[b]:] = 2
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    got = [s.logits()]
    s.step(1, use_graph=False)
    got.append(s.logits())
    s.step(2, use_graph=True)
    got.append(s.logits())
    out = s.output_ids()
    s.close()
    ref_logits, ref_ids = QO.run_model(qmodel, ids, lens, NEW, feed_ids=out[:, S:S + NEW])
    # the GPU feeds back its own greedy ids; the oracle is fed the same ids, so logits are comparable step by step
    # Tolerance: the kernels and the oracle round identically except for fma contraction and summation order; in the
    # SmoothQuant paths a 1-ulp fp16 difference ahead of a quantiser can flip an int8 activation by 1 LSB, which the
    # following GEMMs amplify — so the bound is on the bulk (mean) error, with a looser cap on the worst logit.
    scale = max(np.abs(ref_logits[0]).max(), 1.0)
    sq = mode.startswith('sq')
    for g, r in ((got[0], ref_logits[0]), (got[1], ref_logits[1]), (got[2], ref_logits[3])):
        print(f'[{mode} kv8={int8_kv}] max |d| / scale = {np.abs(g - r).max() / scale:.4g}, mean |d| / scale = {np.abs(g - r).mean() / scale:.4g}')
        # observed on MI355X (r02): SmoothQuant max <= 3.6e-2, mean <= 7.5e-3 of the logit range; weight-only max <= 5.5e-3,
        # mean <= 1.2e-3 - the bounds are those + margin (a wrong per-channel scale on a few columns moves the max past them)
        np.testing.assert_allclose(g, r, atol=(5e-2 if sq else 1e-2) * scale)
        assert np.abs(g - r).mean() < (1e-2 if sq else 2.5e-3) * scale
    # and the quantised model must stay close to its fp16 parent (sanity of the scale algebra, not a kernel check)
    fp = QO.run_fp16_model(cfg, w, ids, lens, NEW, feed_ids=out[:, S:S + NEW])[0]
    tol = {'woq8': 0.15, 'woq4': 1.5, 'sq_static': 0.6, 'sq_static_pc': 0.6, 'sq_dyn': 0.4, 'sq_dyn_pc': 0.4}[mode]
    assert np.abs(got[0] - fp[0]).max() < tol * scale + (0.2 * scale if int8_kv else 0)


@pytest.mark.parametrize('mode,int8_kv', [('fp16', 0), ('sq_static_pc', 1), ('woq4', 1)])
def test_one_layer_at_7b_dimensions_vs_oracle(mode, int8_kv):
    """BASELINE.json's full layer size (D = 4096, 32 heads of 128, I = 11008; the bench configuration is
    sq_static_pc + int8 KV): one decoder layer + head, context of a ragged batch and 3 generation steps (eager and
    graph) against the oracle on identical weights and scales.  Exercises the kernels at exactly the shapes bench.py
    times - 8 KiB GEMV tiles over K = 4096 / 11008, the 256 x 192 / 128 x 128 prefill GEMM tiles need M >= 32 and are
    covered by test_prefill_gemm_every_tile_shape; here the context GEMMs run at M = 2 * 20."""
    cfg, w = synth_model(23, L=1, H=32, D=4096, I=11008, V=512)
    B, S, NEW = 2, 20, 4
    r = np.random.default_rng(9)
    ids = r.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    lens = np.array([S, 13], np.int32)
    for b in range(B):
        ids[b, lens[b]:] = 2
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    got = [s.logits()]
    s.step(1, use_graph=False)
    got.append(s.logits())
    s.step(2, use_graph=True)
    got.append(s.logits())
    out = s.output_ids()
    s.close()
    ref_logits, _ = QO.run_model(qmodel, ids, lens, NEW, feed_ids=out[:, S:S + NEW])
    scale = max(np.abs(ref_logits[0]).max(), 1.0)
    sq = mode.startswith('sq')
    for g, rr in ((got[0], ref_logits[0]), (got[1], ref_logits[1]), (got[2], ref_logits[3])):
        assert np.isfinite(g).all()
        np.testing.assert_allclose(g, rr, atol=(5e-2 if sq else 1e-2) * scale)
        assert np.abs(g - rr).mean() < (1.2e-2 if sq else 2.5e-3) * scale


@pytest.mark.parametrize('geometry', ['small_2_layers', 'one_layer_7b'])
@pytest.mark.parametrize('mode', ['sq_static_pc', 'sq_dyn_pc'])
def test_sq_int8_taps_at_all_four_quantisers_in_lsbs(geometry, mode):
    """SmoothQuant + int8 KV: the int8 operand of each of the layer's four GEMMs (tllm_session_get_tap_ex: behind
    input_layernorm's quantiser, the O-projection's, post_layernorm's and the SwiGLU quantiser) and the layer's fp16 input row in
    every generation step, in LSBs - not through a logit tolerance scaled by the logit range (which would hide a 0.1 logit error
    at scale 2).  Two statements:

    LOCAL (the strong one, static scales): every stage of the layer as a function of the ENGINE'S OWN previous tap equals the
    algorithm's - x -> RMSNorm -> quantiser -> qkv_in;  x, o_in -> O GEMM + residual -> RMSNorm -> quantiser -> mlp_in;
    mlp_in -> fc | gate GEMMs -> SwiGLU -> quantiser -> proj_in;  proj_in -> proj GEMM + residual -> the next layer's x.
    Integers in, integers out: the GEMMs are exact, so the only freedom is an fp32 reduction order ahead of a rounding - at most
    one LSB on well under 1 % of the elements.  (Attention, the one stage that is not a function of taps alone, has its own tests
    at the reference's 2e-3.)

    END TO END against the oracle on the same token prefix: the first layer's taps differ by at most one LSB.  Further down one
    flipped integer moves every output of the next GEMM by a fraction of an fp16 ulp, the next quantiser turns that into flips
    on ~10 % of its elements, and so on: at the second layer of the D = 256 model half of the integers differ by 1 - 4 LSBs
    although every local stage above is exact.  That growth is the quantised model's sensitivity, not an implementation error -
    bench.py's `parity.sq_attribution` measures the same thing at 7B between two torch restatements of the algorithm."""
    if geometry == 'small_2_layers':
        cfg, w = synth_model(11)
        B, S, NEW = 2, 12, 4
        lens = np.array([S, 7], np.int32)
    else:
        cfg, w = synth_model(23, L=1, H=32, D=4096, I=11008, V=512)
        B, S, NEW = 2, 20, 4
        lens = np.array([S, 13], np.int32)
    L = cfg['num_layers']
    r = np.random.default_rng(5)
    ids = r.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    for b in range(B):
        ids[b, lens[b]:] = 2
    qmodel = QO.quantise_model(cfg, w, mode, 1, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode'], debug_taps=1))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    widths = dict(qkv_in=cfg['hidden_size'], o_in=cfg['hidden_size'], mlp_in=cfg['hidden_size'], proj_in=cfg['inter_size'],
                  x_in=cfg['hidden_size'])
    got, got_logits = [], [s.logits()]
    for step in range(NEW - 1):
        s.step(1, use_graph=step >= 1)
        got.append([{n: s.tap(li, n, wd, quantised=True) for n, wd in widths.items()} for li in range(L)])
        got_logits.append(s.logits())
    out = s.output_ids()
    s.close()
    taps = {}
    ref_logits, _ = QO.run_model(qmodel, ids, lens, NEW, feed_ids=out[:, S:S + NEW], taps=taps)
    order = ['qkv_in', 'o_in', 'mlp_in', 'proj_in']
    worst = {}
    for step in range(NEW - 1):
        for li in range(L):
            for n in order:
                d = np.abs(got[step][li][n].astype(np.int32) - taps['gemm_in'][step][li][n].astype(np.int32))
                key = (li, n)
                worst[key] = (max(worst.get(key, (0, 0))[0], int(d.max())), max(worst.get(key, (0, 0))[1], float((d != 0).mean())))
    for li in range(L):
        for n in order:
            mx, frac = worst[(li, n)]
            print(f'[{geometry} {mode}] vs oracle, layer {li} {n}: max {mx} LSB, {100 * frac:.3f} % of the elements differ')
    scale = max(np.abs(ref_logits[0]).max(), 1.0)
    for g, rr in zip(got_logits, ref_logits):
        print(f'[{geometry} {mode}] logits vs oracle: max |d| = {np.abs(g - rr).max():.4g}, mean |d| = {np.abs(g - rr).mean():.4g} (scale {scale:.3g})')
    if mode == 'sq_static_pc':
        eps = cfg['rms_norm_eps']

        def near(name, step, li, want, have):
            d = np.abs(want.astype(np.int32) - have.astype(np.int32))
            print(f'[{geometry} {mode}] local, step {step} layer {li} {name}: max {d.max()} LSB, {100 * (d != 0).mean():.4f} % differ')
            assert d.max() <= 1 and (d != 0).mean() <= 0.005, (name, step, li, d.max(), (d != 0).mean())

        for step in range(NEW - 1):
            for li in range(L):
                lw = qmodel['oracle']['layers'][li]
                t = got[step][li]
                x = t['x_in'].astype(np.float32)
                near('x -> qkv_in', step, li, O.quantize_tensor(O.rmsnorm(x, lw['ln1'], eps), lw['ln1_scale']), t['qkv_in'])
                x1 = O.f16(x + lw['attention.dense'](None, (t['o_in'], None)))
                near('x, o_in -> mlp_in', step, li, O.quantize_tensor(O.rmsnorm(x1, lw['ln2'], eps), lw['ln2_scale']), t['mlp_in'])
                qi = (t['mlp_in'], None)
                near('mlp_in -> proj_in', step, li,
                     O.quantize_tensor(O.swiglu(lw['mlp.fc'](None, qi), lw['mlp.gate'](None, qi)), lw['mlp_qscale']), t['proj_in'])
                if li + 1 < L:  # the residual stream itself: exact GEMM, two fp16 roundings - bit for bit
                    x2 = O.f16(x1 + lw['mlp.proj'](None, (t['proj_in'], None)))
                    np.testing.assert_array_equal(x2.astype(np.float16), got[step][li + 1]['x_in'])
    for n in order:  # the first layer end to end: one LSB at most (two with per-token scales: amax itself may be an ulp apart)
        mx, frac = worst[(0, n)]
        assert mx <= (1 if mode == 'sq_static_pc' else 2) and frac <= (0.005 if n == 'qkv_in' else 0.25), (n, mx, frac)


@pytest.mark.parametrize('mode,int8_kv', [('fp16', 0), ('sq_static_pc', 1)])
def test_batch_invariance_and_session_reuse(mode, int8_kv):
    """Size-independent property: a sequence generates the same tokens alone (batch 1, M = 1 kernels) and inside a ragged
    batch of 8 (M = 8 kernels, padded prompts, masked cache slots), and a session can be set up again for another batch
    shape.  The padded layout shifts positions by (max_input_len - len), which the attention kernels undo exactly
    (MM/...Template.h:1425-1426), so the comparison is token for token on the first steps and within a near-tie
    allowance afterwards."""
    cfg, w = synth_model(41)
    r = np.random.default_rng(17)
    B, S, NEW = 8, 24, 12
    lens = np.array([24, 5, 17, 1, 24, 9, 13, 20], np.int32)
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    batched = s.generate(ids, lens, NEW, end_id=-1)
    single = []
    for b in range(B):
        L = int(lens[b])
        s.setup(1, L, NEW)  # the same session, re-shaped
        single.append(s.generate(ids[b:b + 1, :L], lens[b:b + 1], NEW, end_id=-1)[0, L:L + NEW])
    s.setup(B, S, NEW)
    again = s.generate(ids, lens, NEW, end_id=-1)
    s.close()
    np.testing.assert_array_equal(batched, again)  # re-setup leaves no state behind
    got = batched[:, S:S + NEW]
    single = np.stack(single)
    np.testing.assert_array_equal(got[:, 0], single[:, 0])  # the context phase agrees exactly on the first token
    assert np.mean(got == single) > 0.9, (got, single)


@pytest.mark.parametrize('mode,int8_kv', [('fp16', 0), ('sq_static_pc', 1)])
def test_packed_context_equals_padded(mode, int8_kv):
    """remove_input_padding=1: the context phase runs on the sum(len) real tokens only; logits of the last prompt token,
    the KV cache it leaves behind and the generation that follows must be those of the padded run."""
    cfg, w = synth_model(51)
    r = np.random.default_rng(19)
    B, S, NEW = 4, 40, 8
    lens = np.array([40, 7, 23, 33], np.int32)
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    outs, logits = [], []
    for packed in (0, 1):
        s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode'], remove_input_padding=packed))
        for k, v in qmodel['engine_tensors'].items():
            s.set_tensor(k, v)
        s.finalize()
        s.setup(B, S, NEW)
        s.context(ids, lens)
        logits.append(s.logits())
        s.step(NEW - 1, use_graph=True)
        outs.append(s.output_ids())
        s.close()
    # GEMM rows are computed independently of how many rows there are -> the same bits
    np.testing.assert_array_equal(logits[0], logits[1])
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize('S,int8_kv', [(1080, 1), (1080, 0), (1750, 1), (2200, 0)])
def test_long_contexts_cover_every_generation_attention_geometry(S, int8_kv):
    """The session picks the split-KV geometry of the generation attention from the cache capacity: 12 rows per lane group
    (<= 1536 slots) or 16 (<= 2048) with the merge fused into the O-projection, the fine split with its own combine launch
    beyond.  Context logits and three generation steps (eager + graph) against the oracle at each."""
    cfg, w = synth_model(41)
    B, NEW = 2, 5
    r = np.random.default_rng(S)
    lens = np.array([S, S - 333], np.int32)
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    # weight-only int8 with the int8 cache: the activations stay fp16, so the comparison keeps the tight fp16-path bound (the
    # SmoothQuant paths amplify 1-LSB quantiser flips over a thousand positions into a bulk error that says nothing here)
    mode = 'woq8' if int8_kv else 'fp16'
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids[:, :64], calib_lens=np.array([64, 64], np.int32))
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    got = [s.logits()]
    s.step(1, use_graph=False)
    got.append(s.logits())
    s.step(2, use_graph=True)
    got.append(s.logits())
    out = s.output_ids()
    s.close()
    ref, _ = QO.run_model(qmodel, ids, lens, 4, feed_ids=out[:, S:S + 4])
    scale = max(np.abs(ref[0]).max(), 1.0)
    for g, rr in ((got[0], ref[0]), (got[1], ref[1]), (got[2], ref[3])):
        np.testing.assert_allclose(g, rr, atol=3e-2 * scale)
        assert np.abs(g - rr).mean() < 5e-3 * scale


def test_end_id_stops_a_sequence_and_leaves_the_others_alone():
    """Stop criteria on the device (K/stopCriteriaKernels.cu, generation.py:943-983): once a sequence emits end_id it keeps
    emitting end_id, the other sequences of the batch continue exactly as without a stop token, and generate() returns as
    soon as every sequence has finished."""
    cfg, w = synth_model(61)
    r = np.random.default_rng(23)
    B, S, NEW = 3, 10, 40
    lens = np.array([10, 6, 8], np.int32)
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    s = NativeSession(dict(cfg, quant_mode=0))
    for k, v in w.items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    free = s.generate(ids, lens, NEW, end_id=-1)[:, S:]
    # a token sequence 0 emits at step 5 and that did not occur earlier in that sequence
    k = next(i for i in range(3, NEW) if free[0, i] not in free[0, :i])
    end_id = int(free[0, k])
    s.setup(B, S, NEW)
    stopped = s.generate(ids, lens, NEW, end_id=end_id)[:, S:]
    s.close()
    np.testing.assert_array_equal(stopped[0, :k + 1], free[0, :k + 1])
    assert (stopped[0, k:] == end_id).all()
    for b in (1, 2):
        hit = np.where(free[b] == end_id)[0]
        upto = NEW if len(hit) == 0 else hit[0] + 1
        np.testing.assert_array_equal(stopped[b, :upto], free[b, :upto])
        assert (stopped[b, upto:] == end_id).all()


@pytest.mark.parametrize('mode,int8_kv', [('fp16', 0), ('sq_static_pc', 1), ('woq8', 1)])
def test_more_than_eight_sequences_run_in_slabs_of_eight(mode, int8_kv):
    """batch x beam width beyond 8: the generation GEMVs take 8 rows per launch, the session runs the rows in slabs (VERDICT r2 weak
    #8: build.py's default --max_batch_size 8 with a beam width > 1 was refused).  12 greedy sequences of ragged length against
    the same sequences run alone, and 5 prompts x 2 beams against each prompt searched alone: the slabs must not leak into each
    other (row offsets of activations, residuals, split-KV partials, per-token scales)."""
    cfg, w = synth_model(43)
    r = np.random.default_rng(19)
    B, S, NEW = 12, 20, 8
    lens = r.integers(3, S + 1, B).astype(np.int32)
    lens[0] = S
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    got = s.generate(ids, lens, NEW, end_id=-1)[:, S:S + NEW]
    again = s.generate(ids, lens, NEW, end_id=-1)[:, S:S + NEW]  # graph replay after the first call's capture
    np.testing.assert_array_equal(got, again)
    single = []
    for b in range(B):
        L = int(lens[b])
        s.setup(1, L, NEW)
        single.append(s.generate(ids[b:b + 1, :L], lens[b:b + 1], NEW, end_id=-1)[0, L:L + NEW])
    single = np.stack(single)
    np.testing.assert_array_equal(got[:, 0], single[:, 0])
    assert np.mean(got == single) > 0.9, (got, single)
    # every slab is really its own rows: sequence 9 (second slab) differs from sequence 1 (first slab, same slot in its slab)
    assert not np.array_equal(got[1], got[9])
    # beam search, 5 x 2 hypotheses = 10 sequences
    Bb, W = 5, 2
    s.setup(Bb, S, NEW, beam_width=W)
    s.generate(ids[:Bb], lens[:Bb], NEW)
    beams, cum = s.beam_output()
    for b in range(Bb):
        s.setup(1, S, NEW, beam_width=W)
        s.generate(ids[b:b + 1], lens[b:b + 1], NEW)
        b1, c1 = s.beam_output()
        np.testing.assert_allclose(cum[b], c1[0], atol=0.05)
        np.testing.assert_array_equal(beams[b, 0, :S + 2], b1[0, 0, :S + 2])
    s.close()


# (every mode at the 30B extents; at the 65B extents - 16 - 18 s of numpy oracle each - fp16 and the static SmoothQuant form: the
#  per-token quantiser and the weight-only kernels see nothing at 8192 x 22016 that 6656 x 17920 does not show; suite wall time)
@pytest.mark.parametrize('mode,int8_kv,dims', [('fp16', 0, '30b'), ('woq8', 1, '30b'), ('sq_static_pc', 1, '30b'), ('sq_dyn_pc', 1, '30b'),
                                               ('fp16', 0, '65b'), ('sq_static_pc', 1, '65b')])
def test_one_layer_at_larger_llama_dimensions_vs_oracle(mode, int8_kv, dims):
    """The layer dimensions of LLaMA-30B (D 6656, 52 heads, FFN 17920) and 65B (D 8192, 64 heads, FFN 22016) - what `build.py --n_embd
    --n_head --inter_size` of the reference accepts - with a ragged batch of 5 (the GEMV's 8-row bucket; with fp16 activations its rows
    exceed a CU's LDS and go through in slabs, gemv.hip): rows of 13 - 43 KiB, the third activation-vector bucket for the
    down-projection, 52 / 64 heads in the attention launch and its in-launch merge.  Context + 3 generation steps (eager and graph)
    against the oracle on identical weights and scales."""
    H, D, I = {'30b': (52, 6656, 17920), '65b': (64, 8192, 22016)}[dims]
    cfg, w = synth_model(29, L=1, H=H, D=D, I=I, V=512)
    B, S, NEW = 5, 24, 4
    r = np.random.default_rng(13)
    ids = r.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    lens = np.array([S, 13, 24, 7, 19], np.int32)
    for b in range(B):
        ids[b, lens[b]:] = 2
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    got = [s.logits()]
    s.step(1, use_graph=False)
    got.append(s.logits())
    s.step(2, use_graph=True)
    got.append(s.logits())
    out = s.output_ids()
    s.close()
    ref_logits, _ = QO.run_model(qmodel, ids, lens, NEW, feed_ids=out[:, S:S + NEW])
    scale = max(np.abs(ref_logits[0]).max(), 1.0)
    sq = mode.startswith('sq')
    for g, rr in ((got[0], ref_logits[0]), (got[1], ref_logits[1]), (got[2], ref_logits[3])):
        assert np.isfinite(g).all()
        print(f'[{dims} {mode}] max |d| / scale = {np.abs(g - rr).max() / scale:.4g}, mean |d| / scale = {np.abs(g - rr).mean() / scale:.4g}')
        np.testing.assert_allclose(g, rr, atol=(6e-2 if sq else 1e-2) * scale)
        assert np.abs(g - rr).mean() < (1.5e-2 if sq else 2.5e-3) * scale


@pytest.mark.parametrize('mode,int8_kv', [('fp16', 0), ('sq_static_pc', 1)])
def test_odd_vocabulary_size_vs_oracle(mode, int8_kv):
    """A vocabulary that is no multiple of anything (32001-style checkpoints with an added pad token; here 1003): lm_head rows, the fp32
    logits buffer and the device-side arg-max at a ragged size - context + 3 generation steps against the oracle, and the greedy ids."""
    cfg, w = synth_model(17, V=1003)
    B, S, NEW = 3, 10, 4
    r = np.random.default_rng(3)
    ids = r.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    lens = np.array([S, 4, 9], np.int32)
    for b in range(B):
        ids[b, lens[b]:] = 2
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    got = [s.logits()]
    s.step(1, use_graph=False)
    got.append(s.logits())
    s.step(2, use_graph=True)
    got.append(s.logits())
    out = s.output_ids()
    s.close()
    assert got[0].shape == (B, 1003)
    ref_logits, _ = QO.run_model(qmodel, ids, lens, NEW, feed_ids=out[:, S:S + NEW])
    scale = max(np.abs(ref_logits[0]).max(), 1.0)
    sq = mode.startswith('sq')
    for step, (g, rr) in enumerate(((got[0], ref_logits[0]), (got[1], ref_logits[1]), (got[2], ref_logits[3]))):
        np.testing.assert_allclose(g, rr, atol=(5e-2 if sq else 1e-2) * scale)
    # the sampler's pick is the arg-max of the logits it was given, for every row, also in the last (ragged) stretch of the vocabulary
    np.testing.assert_array_equal(out[:, S], got[0].argmax(-1))


@pytest.mark.parametrize('mode,int8_kv', [('fp16', 0), ('woq8', 1), ('woq4', 0), ('sq_static_pc', 1), ('sq_dyn', 0)])
def test_awkward_dimensions_vs_oracle(mode, int8_kv):
    """Dimensions no production kernel is tiled for - 5 heads of 64 (D = 320), FFN 992, vocabulary 777, a ragged batch of 3 with a
    one-token prompt in it - so that the fall-back paths run: prefill GEMMs whose K is no multiple of the 128-byte K-tile (320 int8
    bytes, 992) or of the weight-only kernel's 64 elements (992), row counts that are no multiple of any tile, the generic context
    attention for 5 heads.  Context (M = 42 rows) + 3 generation steps against the oracle."""
    cfg, w = synth_model(31, L=2, H=5, D=320, I=992, V=777)
    B, S, NEW = 3, 14, 4
    r = np.random.default_rng(7)
    ids = r.integers(3, cfg['vocab_size'], (B, S)).astype(np.int32)
    lens = np.array([S, 1, 9], np.int32)
    for b in range(B):
        ids[b, lens[b]:] = 2
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    got = [s.logits()]
    s.step(1, use_graph=False)
    got.append(s.logits())
    s.step(2, use_graph=True)
    got.append(s.logits())
    out = s.output_ids()
    s.close()
    ref_logits, _ = QO.run_model(qmodel, ids, lens, NEW, feed_ids=out[:, S:S + NEW])
    scale = max(np.abs(ref_logits[0]).max(), 1.0)
    sq = mode.startswith('sq')
    for g, rr in ((got[0], ref_logits[0]), (got[1], ref_logits[1]), (got[2], ref_logits[3])):
        assert np.isfinite(g).all()
        np.testing.assert_allclose(g, rr, atol=(6e-2 if sq else 1e-2) * scale)


def test_a_tensor_parallel_rank_without_its_collectives_runs_alone():
    """Session key no_comm = 1 (bench.py's `tp_rank_prediction`): one rank of a tensor-parallel group - sharded heads, FFN columns and
    vocabulary - runs its launches WITHOUT a communicator, every collective skipped.  Timing only (the hidden states are one rank's
    partial sums), so what is checked is that it finalises, steps eagerly and from the graph, stays finite, and that the SAME shard with
    collectives required refuses to finalise without a communicator."""
    cfg, w = synth_model(5, L=2, H=4, D=256, I=512, V=512)
    tp = 2
    H, D, I, V = cfg['num_heads'], cfg['hidden_size'], cfg['inter_size'], cfg['vocab_size']
    Dh = D // H

    def shard(no_comm):
        s = NativeSession(dict(cfg, quant_mode=0, tp_size=tp, tp_rank=0, no_comm=no_comm))
        for k, v in w.items():
            if k.endswith('attention.qkv.weight'):
                v = v.reshape(3, H, Dh, D)[:, :H // tp].reshape(3 * D // tp, D)
            elif k.endswith('attention.dense.weight'):
                v = v[:, :D // tp]
            elif k.endswith('mlp.fc.weight') or k.endswith('mlp.gate.weight'):
                v = v[:I // tp]
            elif k.endswith('mlp.proj.weight'):
                v = v[:, :I // tp]
            elif k == 'lm_head.weight':
                v = v[:V // tp]
            s.set_tensor(k, np.ascontiguousarray(v))
        return s

    s = shard(1)
    s.finalize()
    s.setup(1, 16, 8)
    s.fake_context(16, seed=3)
    s.step(2, use_graph=False)
    s.step(4, use_graph=True)
    assert np.isfinite(s.output_ids()).all()
    s.close()
    s2 = shard(0)
    with pytest.raises(RuntimeError, match='communicator'):
        s2.finalize()
    s2.close()
