"""Quantised LLaMA paths on the CPU (numpy) — TEST INFRASTRUCTURE ONLY (see llama_oracle.py header).

Restates, on top of llama_oracle.py:
  * weight-only int8/int4 layers: PY/quantization/layer.py:268-382 + Q/quant.py:52-75 (every Linear except lm_head);
  * the SmoothQuant LLaMA block.  The reference's own SmoothQuant-LLaMA never ran (README.md:802-809,856) and is
    structurally wrong as wired (SURVEY.md "fact 1"), so this follows SURVEY Appendix A.4: the working upstream
    GPT-2 pattern (PY/quantization/layer.py:385-439 MLP, :596-852 attention, :223-265 norm+quant) transplanted onto
    the fp16 LLaMA layer (Q/llama_model.py:78-119):
        x -[RMSNorm+quant]-> i8 -[SQ QKV]-> fp16 -[attention, RoPE, KV]-> fp16 -[quant]-> i8 -[SQ O]-> fp16 (+res)
          -[RMSNorm+quant]-> i8 -[SQ fc | SQ gate]-> fp16 -> silu(fc)*gate -[quant]-> i8 -[SQ proj]-> fp16 (+res)
    PARITY UNPINNED above kernel level: there is no reference output for this block; kernel-level semantics are
    pinned (SQ GEMM / quantisers exact vs the reference tests' formulas) and the model is checked against its own
    fp16 parent.
  * scale algebra: Q/convert.py:27-103 (generate_int8) consumption rules of Q/weight_quant.py:116-147, :439-446;
    smoothing Q/smoothquant.py:37-67 folded into the preceding RMSNorm weight (what smooth_gemm's
    `layernorm_weights` argument does for LayerNorm).
"""
import numpy as np

from . import llama_oracle as O

F32 = np.float32
QM = dict(INT4_WEIGHTS=1, INT8_WEIGHTS=2, ACTIVATIONS=4, PER_CHANNEL=8, PER_TOKEN=16, INT8_KV_CACHE=32)

LINEARS = ('attention.qkv', 'attention.dense', 'mlp.fc', 'mlp.gate', 'mlp.proj')


def process_woq_layout(q_kn, bits):
    """numpy restatement of the processed weight-only layout (csrc/kernels/weight_layout.h)."""
    k, n = q_kn.shape
    if bits == 8:
        ldw = (k + 15) // 16 * 16
        out = np.full((n, ldw), 128, np.uint8)
        out[:, :k] = (q_kn.T.astype(np.int16) + 128).astype(np.uint8)
        return out.view(np.int8)
    kp = (k + 31) // 32 * 32
    nib = np.full((n, kp), 8, np.uint8)
    nib[:, :k] = (q_kn.T.astype(np.int16) + 8).astype(np.uint8)
    nib = nib.reshape(n, kp // 8, 8)
    elem_of_nibble = [0, 2, 4, 6, 1, 3, 5, 7]
    word = np.zeros((n, kp // 8), np.uint32)
    for pos, e in enumerate(elem_of_nibble):
        word |= nib[:, :, e].astype(np.uint32) << (4 * pos)
    return word.view(np.uint8).reshape(n, kp // 2).view(np.int8)


class _Linear:
    """One GEMM of the layer in its quantised form; __call__ consumes fp16 activations [M, K]."""

    def __init__(self, kind, **kw):
        self.kind = kind
        self.__dict__.update(kw)

    def __call__(self, x16, in_q=None):
        if self.kind == 'fp16':
            return O.gemm_fp16(x16, self.w)
        if self.kind == 'woq':
            return O.woq_matmul(x16, self.q_kn, self.scales, reference_rounding=getattr(self, 'reference_rounding', False)
                                and x16.shape[0] <= 8)
        # SmoothQuant: `in_q` = (int8 activations, per-token scales or None)
        xq, tok = in_q
        if tok is None:
            return O.sq_gemm(xq, self.w_i8, self.act_scale, self.per_channel_scale)
        return O.sq_gemm(xq, self.w_i8, tok, self.per_channel_scale)


def _forward(model, ids, lens, n_new, feed_ids=None, capture=None, taps=None, reference_rounding=False, start_caches=None):
    """Context step + n_new-1 generation steps.  Returns ([logits per step], greedy ids [B, n_new]).
    `start_caches` (one [B, 2, H, S + n_new, Dh] array per layer, with `feed_ids`): the context step is skipped and generation
    starts from these cache contents at length S - parity tests of the generation kernels at long contexts seed both sides with
    the same cache bytes instead of running two prefills; logits[0] is then None.
    `taps` (a dict) receives 'caches' (the per-layer KV caches, live objects: their state after the last step),
    'caches_after_context' (copies), 'attn_ctx' = [step][layer] attention output [B, H*Dh] of every generation step
    (the O-projection's input before its quantiser) and 'gemm_in' = [step][layer] dict(qkv_in, o_in, mlp_in, proj_in): the
    operand of each of the layer's four GEMMs in that generation step - int8 behind its quantiser for SmoothQuant
    (K/quantization.cu:31-118), the fp16 activation otherwise.  `reference_rounding`: see llama_oracle.mmha_decode / woq_matmul."""
    cfg = model['cfg']
    B, S = ids.shape
    H, D = cfg['num_heads'], cfg['hidden_size']
    Dh = D // H
    smax = S + n_new
    L = cfg['num_layers']
    eps = cfg.get('rms_norm_eps', 1e-6)
    int8_kv = model['int8_kv']
    caches = [np.zeros((B, 2, H, smax, Dh), np.int8 if int8_kv else np.float16) for _ in range(L)]
    sq, per_token = model['sq'], model['per_token']

    def cap(name, x=None, y=None):
        if capture is None:
            return
        d = capture.setdefault(name, {})
        if x is not None:
            ax = np.abs(x.reshape(-1, x.shape[-1])).max(0)
            d['x'] = np.maximum(d.get('x', 0), ax)
        if y is not None:
            ay = np.abs(y.reshape(-1, y.shape[-1])).max(0)
            d['y'] = np.maximum(d.get('y', 0), ay)

    def quant_in(x16, static_scale):
        if not sq:
            return None
        if per_token:
            return O.quantize_per_token(x16)
        return O.quantize_tensor(x16, static_scale), None

    def layer(li, x16, rows_valid, attn_fn, gemm_in=None):
        lw = model['layers'][li]
        M = x16.shape[0]
        rec = {}

        def operand(name, x_f16, q):  # what the GEMM consumes: the quantiser's int8, or the fp16 activation
            rec[name] = (q[0] if q is not None else x_f16).copy()
            return q

        h = O.rmsnorm(x16, lw['ln1'], eps)
        cap(f'{li}.attention.qkv', x=h[rows_valid])
        qkv = lw['attention.qkv'](h, operand('qkv_in', h, quant_in(h, lw.get('ln1_scale'))))
        cap(f'{li}.attention.qkv', y=qkv[rows_valid])
        ctx = attn_fn(qkv, caches[li], lw)
        cap(f'{li}.attention.dense', x=ctx[rows_valid])
        attn = lw['attention.dense'](ctx, operand('o_in', ctx, quant_in(ctx, lw.get('attn_qscale'))))
        cap(f'{li}.attention.dense', y=attn[rows_valid])
        x1 = O.f16(x16 + attn)
        h2 = O.rmsnorm(x1, lw['ln2'], eps)
        cap(f'{li}.mlp.fc', x=h2[rows_valid])
        qi = operand('mlp_in', h2, quant_in(h2, lw.get('ln2_scale')))
        g = lw['mlp.fc'](h2, qi)
        u = lw['mlp.gate'](h2, qi)
        cap(f'{li}.mlp.fc', y=g[rows_valid])
        cap(f'{li}.mlp.gate', y=u[rows_valid])
        inter = O.swiglu(g, u)
        cap(f'{li}.mlp.proj', x=inter[rows_valid])
        m = lw['mlp.proj'](inter, operand('proj_in', inter, quant_in(inter, lw.get('mlp_qscale'))))
        cap(f'{li}.mlp.proj', y=m[rows_valid])
        if gemm_in is not None:
            gemm_in.append(rec)
        return O.f16(x1 + m)

    if start_caches is not None:
        assert feed_ids is not None and len(start_caches) == L
        caches = [np.array(c, copy=True) for c in start_caches]
        assert all(c.shape == (B, 2, H, smax, Dh) for c in caches)
    # ---- context
    x = O.f16(model['emb'][ids]).reshape(B * S, D)
    valid = np.concatenate([np.arange(S) < lens[b] for b in range(B)])

    def ctx_attn(qkv, cache, lw):
        out, _ = O.context_attention(qkv.reshape(B, S, 3 * D), cache, lens, H, Dh, Dh, True, 1.0, lw.get('kv_oq'))
        return out.reshape(B * S, D)

    for li in range(L if start_caches is None else 0):
        x = layer(li, x, valid, ctx_attn)
    if taps is not None:
        taps['caches'] = caches
        taps['caches_after_context'] = [c.copy() for c in caches]
        taps['attn_ctx'] = []
        taps['gemm_in'] = []
    x = x.reshape(B, S, D)
    last = np.stack([x[b, int(lens[b]) - 1] for b in range(B)])
    logits = [(O.rmsnorm(last, model['lnf'], eps) @ model['head'].T).astype(F32)] if start_caches is None else [None]
    gen = [logits[0].argmax(-1) if start_caches is None else np.zeros(B, np.int64)]
    masked = np.zeros((B, smax), np.int32)
    for b in range(B):
        masked[b, lens[b]:S] = 1
    # ---- generation
    for step in range(n_new - 1):
        cur = feed_ids[:, step] if feed_ids is not None else gen[-1]
        xs = O.f16(model['emb'][cur])
        tl = S + step

        if taps is not None:
            taps['attn_ctx'].append([])
            taps['gemm_in'].append([])

        def dec_attn(qkv, cache, lw):
            c = O.mmha_decode(qkv, cache, [tl] * B, lens, S, tl, H, Dh, Dh, True, 1.0, masked, lw.get('kv_oq'),
                              lw.get('kv_qo'), reference_rounding=reference_rounding)
            if taps is not None:
                taps['attn_ctx'][-1].append(c.copy())
            return c

        for li in range(L):
            xs = layer(li, xs, np.ones(B, bool), dec_attn, taps['gemm_in'][-1] if taps is not None else None)
        logits.append((O.rmsnorm(xs, model['lnf'], eps) @ model['head'].T).astype(F32))
        gen.append(logits[-1].argmax(-1))
    return logits, np.stack(gen, 1)


def _fp16_model(cfg, w):
    f = lambda k: np.asarray(w[k], dtype=F32)  # (no copy when the caller already holds fp32: the 65B-dimension tests are copy-bound)
    m = dict(cfg=cfg, sq=False, per_token=False, int8_kv=False, emb=f('vocab_embedding.weight'), lnf=f('ln_f.weight'),
             head=f('lm_head.weight'), layers=[])
    for i in range(cfg['num_layers']):
        p = f'layers.{i}.'
        lw = dict(ln1=f(p + 'input_layernorm.weight'), ln2=f(p + 'post_layernorm.weight'))
        for n in LINEARS:
            lw[n] = _Linear('fp16', w=f(p + n + '.weight'))
        m['layers'].append(lw)
    return m


def run_fp16_model(cfg, w, ids, lens, n_new, feed_ids=None):
    return _forward(_fp16_model(cfg, w), ids, lens, n_new, feed_ids)


def run_model(qmodel, ids, lens, n_new, feed_ids=None, taps=None, reference_rounding=False, start_caches=None):
    m = qmodel['oracle']
    for lw in m['layers']:
        for n in LINEARS:
            if lw[n].kind == 'woq':
                lw[n].reference_rounding = reference_rounding
    return _forward(m, ids, lens, n_new, feed_ids, taps=taps, reference_rounding=reference_rounding, start_caches=start_caches)


def quantise_model(cfg, w, mode, int8_kv, calib_ids, calib_lens, alpha=0.5):
    """mode in {fp16, woq8, woq4, sq_static, sq_static_pc, sq_dyn, sq_dyn_pc}.  Returns
    {'quant_mode', 'engine_tensors' (numpy, engine naming/layouts), 'oracle' (model for run_model)}."""
    L = cfg['num_layers']
    sq = mode.startswith('sq')
    woq_bits = {'woq8': 8, 'woq4': 4}.get(mode)
    per_token = 'dyn' in mode
    per_channel = mode.endswith('_pc')
    et = {k: w[k] for k in ('vocab_embedding.weight', 'ln_f.weight', 'lm_head.weight')}  # lm_head stays fp16
    om = None  # filled from the (smoothed) fp16 model below
    qm = 0
    if woq_bits:
        qm = QM['INT4_WEIGHTS'] if woq_bits == 4 else QM['INT8_WEIGHTS']
    if sq:
        qm = QM['INT8_WEIGHTS'] | QM['ACTIVATIONS'] | (QM['PER_TOKEN'] if per_token else 0) | \
            (QM['PER_CHANNEL'] if per_channel else 0)
    if int8_kv:
        qm |= QM['INT8_KV_CACHE']

    # ---- SmoothQuant: smooth (qkv with ln1, fc|gate with ln2), then calibrate the smoothed fp16 model
    work = {k: v.astype(F32) for k, v in w.items()}
    if sq:
        cap0 = {}
        _forward(_fp16_model(cfg, work), calib_ids, calib_lens, 1, capture=cap0)
        for i in range(L):
            p = f'layers.{i}.'
            (wq, ), s = O.smooth_gemm([work[p + 'attention.qkv.weight']], cap0[f'{i}.attention.qkv']['x'], alpha)
            work[p + 'attention.qkv.weight'] = O.f16(wq)
            work[p + 'input_layernorm.weight'] = O.f16(work[p + 'input_layernorm.weight'] / s)
            (wf, wg), s2 = O.smooth_gemm([work[p + 'mlp.fc.weight'], work[p + 'mlp.gate.weight']],
                                         cap0[f'{i}.mlp.fc']['x'], alpha)
            work[p + 'mlp.fc.weight'], work[p + 'mlp.gate.weight'] = O.f16(wf), O.f16(wg)
            work[p + 'post_layernorm.weight'] = O.f16(work[p + 'post_layernorm.weight'] / s2)
    smoothed = _fp16_model(cfg, work)
    om = dict(smoothed, sq=sq, per_token=per_token, int8_kv=bool(int8_kv), layers=[])
    cap = {}
    if sq or int8_kv:  # activation ranges: the static SmoothQuant scales and the KV-cache scale (nothing else reads them)
        _forward(smoothed, calib_ids, calib_lens, 1, capture=cap)

    for i in range(L):
        p = f'layers.{i}.'
        lw = dict(ln1=work[p + 'input_layernorm.weight'], ln2=work[p + 'post_layernorm.weight'])
        et[p + 'input_layernorm.weight'] = lw['ln1'].astype(np.float16)
        et[p + 'post_layernorm.weight'] = lw['ln2'].astype(np.float16)
        for n in LINEARS:
            W = work[p + n + '.weight']  # [N, K]
            if woq_bits:
                q, s = O.woq_quantize(W.T, woq_bits)
                lw[n] = _Linear('woq', q_kn=q, scales=s)
                et[p + n + '.weight'] = process_woq_layout(q, woq_bits)
                et[p + n + '.per_channel_scale'] = s.astype(np.float16)
            elif sq:
                cname = f'{i}.{n}' if n != 'mlp.gate' else f'{i}.mlp.fc'  # gate shares fc's input
                x_max = F32(cap[cname]['x'].max())
                y_max = F32(cap[f'{i}.{n}']['y'].max())
                if per_channel:
                    s_w = (F32(127.0) / np.abs(W).max(axis=1)).astype(F32)  # per output channel
                else:
                    s_w = np.array([F32(127.0) / np.abs(W).max()], F32)
                w_i8 = np.clip(np.round(W * (s_w[:, None] if per_channel else s_w)), -127, 127).astype(np.int8)
                s_x = F32(127.0) / x_max
                if per_token:
                    pcs = (F32(1.0) / s_w).astype(F32)  # scale_w_quant_orig[.col]
                    lw[n] = _Linear('sq', w_i8=w_i8, per_channel_scale=pcs)
                else:
                    s_y = F32(127.0) / y_max
                    pcs = (s_y / (s_x * s_w)).astype(F32)  # scale_y_accum_quant[.col]
                    act = np.array([[y_max / F32(127.0)]], F32)  # scale_y_quant_orig
                    lw[n] = _Linear('sq', w_i8=w_i8, per_channel_scale=pcs, act_scale=act)
                    et[p + n + '.act_scale'] = act
                    key = {'attention.qkv': 'ln1_scale', 'mlp.fc': 'ln2_scale', 'attention.dense': 'attn_qscale',
                           'mlp.proj': 'mlp_qscale'}.get(n)
                    if key:
                        lw[key] = s_x
                        ename = {'ln1_scale': 'input_layernorm.scale_to_int', 'ln2_scale': 'post_layernorm.scale_to_int',
                                 'attn_qscale': 'attention.quantization_scaling_factor',
                                 'mlp_qscale': 'mlp.quantization_scaling_factor'}[key]
                        et[p + ename] = np.array([s_x], F32)
                et[p + n + '.weight'] = w_i8
                et[p + n + '.per_channel_scale'] = pcs.reshape(1, -1)
            else:
                lw[n] = _Linear('fp16', w=W)
                et[p + n + '.weight'] = W.astype(np.float16)
        if int8_kv:
            # kv_quant_orig = scale_y_quant_orig of the QKV output, kv_orig_quant = 1 / that (Q/weight_quant.py:439-446)
            y_max = F32(cap[f'{i}.attention.qkv']['y'].max())
            lw['kv_qo'] = y_max / F32(127.0)
            lw['kv_oq'] = F32(1.0) / lw['kv_qo']
            et[p + 'attention.kv_orig_quant_scale'] = np.array([lw['kv_oq']], F32)
            et[p + 'attention.kv_quant_orig_scale'] = np.array([lw['kv_qo']], F32)
        om['layers'].append(lw)
    return dict(quant_mode=qm, engine_tensors=et, oracle=om)
