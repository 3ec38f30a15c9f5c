"""The accuracy half of BASELINE.json's metric ("... ; ROUGE-L delta vs HF", logits within a stated fp16 tolerance) at LLaMA-7B,
and BASELINE.json configs[0] (HF transformers on the host CPU, batch 1, prompt 128) - called by bench.py, which puts the
result into its JSON line as `parity` (and reuses the CPU model for `cpu_baseline`).

No checkpoint, dataset or tokenizer exists on the GPU box, so this is the substitute SURVEY.md section 8d specifies:

  * ONE seeded fp16 parent model of the 7B architecture (Xavier-uniform matrices as tensorrt_llm.Parameter draws them,
    PY/parameter.py:28-38; RMSNorm weights 1 + 0.1 U(-1, 1); embeddings N(0, 0.02); 1 % of the hidden channels carry
    x20 outliers so that SmoothQuant has something to smooth), built ON THE GPU as an HF LlamaForCausalLM;
  * the reference path = HF transformers fp32 on identical weights.  The long runs (N prompts x 128 new tokens; the
    reference's own check is 20 articles x 100 tokens, Q/summarize.py:91,352) use HF fp32 ON THE GPU through torch - exactly what
    the reference's run_hf.py / summarize.py do (Q/run_hf.py:55-57 `.cuda()`; torch here is the checker, never the product
    path).  The host-CPU fp32 model (BASELINE.json configs[0]: batch 1, prompt 128, 16 new tokens) is timed for `cpu_baseline`
    and cross-checks the GPU generator on the first prompt;
  * the product engines from that parent through the product's own converter (examples/llama_quant/inmemory.py ->
    smoothquant.capture_activation_range, hf_llama_convert.smooth_llama_model, convert.generate_int8,
    tllm_symmetric_quantize_last_axis): fp16, weight-only int8 + int8 KV, SmoothQuant per-channel static int8 + int8 KV
    (BASELINE.json configs[1..3]) and the per-token SmoothQuant flavour;
  * a torch restatement of the SmoothQuant-static engine's ALGORITHM on the same int8 weights and scales (`FakeQuantSQ` below:
    exact integer GEMMs, the reference's quantiser / epilogue / int8-KV rounding points, everything else fp32), so that the
    distance engine <-> HF splits into  HIP kernels <-> algorithm  and  algorithm <-> HF.

Per engine, aggregated over the prompts: max |logit error| at steps 0 / 1 / 64 and over all steps against HF evaluated ON THE
ENGINE'S OWN TOKEN PATH (one teacher-forced forward per prompt), so that a flipped near-tie does not turn every later step into a
comparison of two different sentences; arg-max agreement on that path; free-running token match and in-repo ROUGE-L
(summarize.py) of the generated token strings against HF's free-running greedy output and its delta to the fp16 engine's."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
EX = os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant')


def _paths():
    for p in (os.path.join(ROOT, 'trtllm-llama_amd'), EX):
        if p not in sys.path:
            sys.path.insert(0, p)


def llama_config(layers, transformers):
    # no EOS / BOS / PAD ids: generation never stops early (EOS stopping disabled)
    return transformers.LlamaConfig(hidden_size=4096, num_attention_heads=32, num_key_value_heads=32, intermediate_size=11008,
                                    vocab_size=32000, num_hidden_layers=layers, max_position_embeddings=2048, rms_norm_eps=1e-6,
                                    attention_bias=False, tie_word_embeddings=False, eos_token_id=None, bos_token_id=None,
                                    pad_token_id=None)


def build_parent(torch, dev, layers, seed=0):
    """The seeded fp16 parent as an HF model on `dev`."""
    import transformers
    cfg = llama_config(layers, transformers)
    with torch.device('meta'):
        model = transformers.LlamaForCausalLM(cfg)
    model = model.to_empty(device=dev).half().eval()
    g = torch.Generator(device=dev).manual_seed(seed)
    D = cfg.hidden_size
    n_out = max(1, D // 100)
    outlier = torch.randperm(D, generator=g, device=dev)[:n_out]
    with torch.no_grad():
        for name, p in model.named_parameters():
            if 'norm' in name:
                p.copy_((1 + 0.1 * (torch.rand(p.shape, generator=g, device=dev) * 2 - 1)).half())
            elif 'embed_tokens' in name:
                e = torch.randn(p.shape, generator=g, device=dev) * 0.02
                e[:, outlier] *= 20.0
                p.copy_(e.half())
            else:
                n, k = p.shape
                r = (6.0 / (n + k)) ** 0.5
                w = (torch.rand(p.shape, generator=g, device=dev) * 2 - 1) * r
                if name.endswith('o_proj.weight') or name.endswith('down_proj.weight'):
                    w[outlier, :] *= 6.0  # the writers into the residual stream keep the outlier channels alive with depth
                p.copy_(w.half())
        if hasattr(model.model, 'rotary_emb'):  # buffers are not parameters: to_empty left them uninitialised
            model.model.rotary_emb = type(model.model.rotary_emb)(config=cfg).to(dev)
    return model, cfg


def to_fp32(torch, parent, cfg, device):
    """HF fp32 model on `device` holding exactly the parent's (fp16-representable) weights."""
    import transformers
    with torch.device('meta'):
        m = transformers.LlamaForCausalLM(cfg)
    m = m.to_empty(device=device).float().eval()
    src = dict(parent.named_parameters())
    with torch.no_grad():
        for name, p in m.named_parameters():
            p.copy_(src[name].detach().float().to(device))
        if hasattr(m.model, 'rotary_emb'):
            m.model.rotary_emb = type(m.model.rotary_emb)(config=cfg).to(device)
    return m


def to_cpu_fp32(torch, parent, cfg):
    return to_fp32(torch, parent, cfg, 'cpu')


def rouge_l_ids(pred_ids, ref_ids):
    """ROUGE-L F-measure x 100 of two token-id sequences through the product's own implementation (summarize.py)."""
    import summarize
    return 100.0 * summarize.rouge_l(' '.join(str(int(t)) for t in pred_ids), ' '.join(str(int(t)) for t in ref_ids))


QM = dict(fp16=0, woq8=2, sq=2 | 4 | 8, sq_dyn=2 | 4 | 8 | 16)
MODES = ('fp16', 'woq8', 'sq', 'sq_dyn')
INT8_KV = 32


# ------------------------------------------------------------------------------------------------------------------
# The SmoothQuant-static + int8-KV engine's algorithm in torch (checker; attribution of the engine <-> HF distance).
# Same tensors the engine was given (int8 weights [N, K], per-channel accumulator scales, static activation scales, KV scales);
# rounding points of oracle/quant_oracle.py::_forward / oracle/llama_oracle.py (which cite the reference lines they restate):
#   RMSNorm: fp32 statistics, normalise -> fp16 -> * gamma -> fp16            (PY/functional.py:3195-3219)
#   quantiser: sat(rni(float(x16) * s))                                       (K/quantization.cu:31-64)
#   GEMM: exact int32 accumulate, fp16(float(acc) * (s_col * s_row))          (epilogue_per_row_per_col_scale.h:279-347)
#   RoPE fp32 -> fp16; prompt rows attend to fp16 K / V (context phase, K/unfusedAttentionKernels.cu:205-258: fp32 softmax ->
#   fp16 probabilities); generated rows attend to the int8 cache for every earlier position - fp16(float(q8) * s^-1) - and to
#   their own un-quantised k, v (MM/...Template.h:1490-1549, :1719-1779: softmax = exp(qk - max) / (sum + 1e-6) -> fp16)
#   SwiGLU: silu -> fp16 -> * up -> fp16                                      (PY/layers/mlp.py:68-73)
# Reductions run in float64 and are rounded once (no claim about any kernel's summation order).  `reduce_dtype=torch.float32`
# is the CONTROL: the same algorithm, the same integers, only the floating-point reductions (RMSNorm statistics, QK^T, softmax
# sums, PV, lm_head) accumulate in fp32 in torch's order instead - two correct restatements that differ in nothing but
# summation order.  Their distance is what one flipped int8 (an fp16 value one ulp apart ahead of a quantiser) grows into
# through 32 quantised layers, i.e. the floor below which "engine vs algorithm" cannot say anything about the kernels.
# ------------------------------------------------------------------------------------------------------------------
class FakeQuantSQ:

    def __init__(self, torch, tensors, layers, heads=32, eps=1e-6, reduce_dtype=None, as_built=False):
        self.t, self.torch, self.L, self.H, self.eps = tensors, torch, layers, heads, eps
        self.rd = reduce_dtype or torch.float64
        # as_built: the two places where the HIP attention rounds LATER than the reference ("Where the HIP
        # kernels round later than the reference, on purpose"): cached K / V are used as exact integers times the scale (the
        # reference rounds every dequantised element to fp16 first, MM/...Utils.h:2358-2365), and the probabilities are rounded to
        # fp16 UN-normalised, the division by the row sum coming once at the end (the reference rounds p / sum).  With the flag
        # the restatement takes the engine's rounding points, so engine-vs-restatement measures summation order only.
        self.as_built = as_built

    def f16(self, x):
        return x.half().float()

    def rms(self, x, gamma):
        var = (x.to(self.rd) * x.to(self.rd)).mean(-1, keepdim=True)
        inv = (1.0 / (var + self.eps).sqrt()).float()
        return self.f16(self.f16(x * inv) * gamma.float())

    def quant(self, x16, scale):
        return (x16 * scale.float().reshape(())).round().clamp_(-128, 127)  # round-half-even = rni; values are finite

    def gemm(self, q, prefix):
        t = self.t
        acc = (q.double() @ t[prefix + '.weight'].double().t())  # exact: |sum| < 2^53
        s = (t[prefix + '.per_channel_scale'].float().reshape(1, -1) * t[prefix + '.act_scale'].float().reshape(1, 1))
        return self.f16(acc.float() * s)

    def rope(self, x, pos):  # x [T, H, Dh] fp32 holding fp16; NeoX pairing (j, j + Dh/2)
        torch = self.torch
        Dh = x.shape[-1]
        j = torch.arange(Dh // 2, device=x.device, dtype=torch.float64)
        ang = pos.double()[:, None] / torch.pow(torch.tensor(10000.0, dtype=torch.float64, device=x.device), 2.0 * j / Dh)
        c, s = ang.cos().float()[:, None, :], ang.sin().float()[:, None, :]
        a, b = x[..., :Dh // 2], x[..., Dh // 2:]
        return self.f16(torch.cat([c * a - s * b, c * b + s * a], -1))

    def attention(self, qkv, P, kv_oq, kv_qo):
        """qkv [T, 3 * D] (one sequence, T = prompt + generated); rows < P are context-phase rows, rows >= P generation steps."""
        torch = self.torch
        T, H = qkv.shape[0], self.H
        Dh = qkv.shape[1] // 3 // H
        q, k, v = (qkv[:, i * H * Dh:(i + 1) * H * Dh].reshape(T, H, Dh) for i in range(3))
        pos = torch.arange(T, device=qkv.device)
        q, k = self.rope(q, pos), self.rope(k, pos)
        # what a generation step reads back from the int8 cache
        k8 = (k * kv_oq).round().clamp_(-128, 127) * kv_qo
        v8 = (v * kv_oq).round().clamp_(-128, 127) * kv_qo
        if not self.as_built:
            k8, v8 = self.f16(k8), self.f16(v8)
        inv = 1.0 / (Dh ** 0.5)
        qh, kh, vh, k8h, v8h = (z.permute(1, 0, 2).to(self.rd) for z in (q, k, v, k8, v8))  # [H, T, Dh]
        s_f = (qh @ kh.transpose(1, 2)).float() * inv    # fp16 keys
        s_q = (qh @ k8h.transpose(1, 2)).float() * inv   # keys through the cache
        row = torch.arange(T, device=qkv.device)[:, None]
        col = torch.arange(T, device=qkv.device)[None, :]
        gen = (row >= P)
        use_q = gen & (col < row)
        sc = torch.where(use_q[None], s_q, s_f)
        sc = sc.masked_fill((col > row)[None], float('-inf'))
        mx = sc.max(-1, keepdim=True).values
        e = (sc - mx).exp()
        ssum = e.to(self.rd).sum(-1, keepdim=True).float()
        if self.as_built:
            p = self.f16(e).to(self.rd)  # rounded relative to the row maximum; normalised once, behind the P V sum
            norm = torch.where(gen[None], 1.0 / (ssum + 1e-6), 1.0 / ssum)
        else:
            p_ctx = self.f16(e / ssum)
            p_gen = self.f16(e * (1.0 / (ssum + 1e-6)))
            p = torch.where(gen[None], p_gen, p_ctx).to(self.rd)
            norm = None
        p_cache = torch.where(use_q[None], p, torch.zeros_like(p))
        p_own = p - p_cache
        acc = (p_cache @ v8h + p_own @ vh).float()
        out = self.f16(acc * norm if norm is not None else acc)  # [H, T, Dh]
        return out.permute(1, 0, 2).reshape(T, H * Dh)

    def forward(self, ids, P, taps=None, first_row=0):
        """ids int64 [B, T] (prompts + teacher-forced continuations) -> fp32 logits [B, T - first_row, V] of rows >= first_row."""
        torch, t = self.torch, self.t
        B, T = ids.shape
        x = t['vocab_embedding.weight'][ids.reshape(-1)].float()  # [B * T, D]: the GEMMs run on all sequences at once
        for i in range(self.L):
            p = f'layers.{i}.'
            h = self.rms(x, t[p + 'input_layernorm.weight'])
            hq = self.quant(h, t[p + 'input_layernorm.scale_to_int'])
            qkv = self.gemm(hq, p + 'attention.qkv')
            kv_oq = t[p + 'attention.kv_orig_quant_scale'].float().reshape(())
            kv_qo = t[p + 'attention.kv_quant_orig_scale'].float().reshape(())
            ctx = torch.cat([self.attention(qkv[b * T:(b + 1) * T], P, kv_oq, kv_qo) for b in range(B)])
            cq = self.quant(ctx, t[p + 'attention.quantization_scaling_factor'])
            x = self.f16(x + self.gemm(cq, p + 'attention.dense'))
            h2 = self.rms(x, t[p + 'post_layernorm.weight'])
            h2q = self.quant(h2, t[p + 'post_layernorm.scale_to_int'])
            a, b = self.gemm(h2q, p + 'mlp.fc'), self.gemm(h2q, p + 'mlp.gate')
            inter = self.f16(self.f16(a / (1.0 + torch.exp(-a))) * b)
            iq = self.quant(inter, t[p + 'mlp.quantization_scaling_factor'])
            x = self.f16(x + self.gemm(iq, p + 'mlp.proj'))
            if taps is not None:
                taps.append(dict(qkv_in=hq, o_in=cq, mlp_in=h2q, proj_in=iq))
        y = self.rms(x.reshape(B, T, -1)[:, first_row:], t['ln_f.weight'])
        return (y.to(self.rd) @ t['lm_head.weight'].to(self.rd).t()).float()


def _err_stats(np, err, steps):
    """err [prompts, new, vocab] -> the per-step maxima the metric quotes + overall."""
    d = {f'step_{s}': float(err[:, s].max()) for s in steps if s < err.shape[1]}
    d['all_steps'] = float(err.max())
    return d


def run(torch, dev, layers=32, prompt_len=128, new_tokens=128, n_prompts=8, cpu_threads=64, cpu_new_tokens=16, calib_samples=16,
        calib_len=128, cpu_leg=True, log=None):
    """Returns (parity dict, cpu_model, cpu_info).  The caller owns / frees cpu_model."""
    import numpy as np
    _paths()
    import inmemory
    import run_hf
    import smoothquant
    from tensorrt_llm.runtime.native import NativeSession
    log = log or (lambda *a: None)
    t0 = time.perf_counter()
    parent, hf_cfg = build_parent(torch, dev, layers)
    g = torch.Generator().manual_seed(1)
    prompts = torch.randint(3, 32000, (n_prompts, prompt_len), generator=g)  # ids 0-2 reserved, seed 1
    gc = torch.Generator().manual_seed(2)
    calib = [torch.randint(3, 32000, (1, calib_len), generator=gc) for _ in range(calib_samples)]
    act = smoothquant.capture_activation_range(parent, calib, num_samples=calib_samples)
    log(f'parent + calibration: {time.perf_counter() - t0:.1f} s')
    sd = dict(parent.state_dict())
    cfg = dict(num_layers=layers, num_heads=32, hidden_size=4096, inter_size=11008, vocab_size=32000, max_position_embeddings=2048,
               rms_norm_eps=1e-6)
    lens = np.array([prompt_len], np.int32)
    P, N = prompt_len, new_tokens
    gpu = {}
    sq_tensors = None
    for mode in MODES:
        t1 = time.perf_counter()
        int8_kv = mode != 'fp16'
        tensors = inmemory.engine_tensors(sd, layers, mode='sq' if mode.startswith('sq') else mode, act_range=act if int8_kv else None,
                                          per_channel=True, per_token=mode == 'sq_dyn', int8_kv=int8_kv, num_heads=32,
                                          threads=cpu_threads)
        s = NativeSession(dict(cfg, quant_mode=QM[mode] | (INT8_KV if int8_kv else 0)))
        for k, v in tensors.items():
            s.set_tensor(k, v)
        s.finalize()
        s.setup(1, P, N)
        stream = torch.cuda.current_stream().cuda_stream
        all_logits = np.empty((n_prompts, N, cfg['vocab_size']), np.float32)
        all_tokens = np.empty((n_prompts, N), np.int64)
        for pi in range(n_prompts):
            ids_np = prompts[pi:pi + 1].numpy().astype(np.int32)
            s.context(ids_np, lens, stream=stream)
            all_logits[pi, 0] = s.logits(stream=stream)[0]
            for k in range(1, N):
                s.step(1, use_graph=(pi > 0 or k > 1), stream=stream)  # first step eager, the rest replayed from the step's hipGraph
                all_logits[pi, k] = s.logits(stream=stream)[0]
            all_tokens[pi] = s.output_ids(stream=stream)[0, P:P + N]
        s.close()
        if mode == 'sq':
            sq_tensors = tensors
        del tensors
        torch.cuda.empty_cache()
        gpu[mode] = dict(logits=all_logits, tokens=all_tokens)
        log(f'{mode}: converted + {n_prompts} x ({P} + {N}) generated in {time.perf_counter() - t1:.1f} s')

    # ---- the reference path for the long runs: HF fp32 on the GPU, same weights (Q/run_hf.py:55-57, Q/summarize.py:207-216)
    t1 = time.perf_counter()
    ref = to_fp32(torch, parent, hf_cfg, dev)
    with torch.no_grad():
        seq, hf_logits = run_hf.hf_generate(ref, prompts.to(dev), N, eos_token_id=None, pad_token_id=0, return_logits=True)
    hf_tokens = seq[:, P:].cpu().numpy()
    hf_logits = hf_logits.permute(1, 0, 2).cpu().numpy()  # [prompts, new, vocab]
    scale = float(np.abs(hf_logits).max())
    log(f'HF fp32 on the GPU: {n_prompts} x generate({P} + {N}) in {time.perf_counter() - t1:.1f} s')
    steps = (0, 1, 64, N - 1)
    res = {'shape': f'{n_prompts} seeded prompts x (batch 1, prompt {P}, {N} new tokens), greedy, EOS off '
                    f'(BASELINE.json configs[0] prompt shape; SURVEY 8d: 128 new tokens, logit error at steps 0 / 1 / 64)',
           'weights': 'one seeded fp16 LLaMA-7B parent (Xavier, x20 outlier channels); every HF model holds the same values in fp32',
           'reference': 'HF transformers LlamaForCausalLM fp32 on the GPU via run_hf.hf_generate (the reference\'s own run_hf.py / '
                        'summarize.py run HF on the GPU); cross-checked against HF fp32 on the host CPU on prompt 0',
           'layers': layers, 'prompts': n_prompts, 'new_tokens': N, 'logit_scale_max_abs': scale,
           'tolerance': 'reference bound: logits atol 1e-1 (T/tests/model/test_llama.py:286-288, fp16 models); ROUGE-L delta <= 1 '
                        '(README.md:921).  SmoothQuant: see `sq_attribution` and DESIGN.md section 2 for the stated bound',
           'configs': {'fp16': 'fp16 + fp16 KV (BASELINE configs[1])', 'woq8': 'weight-only int8 + int8 KV (configs[2])',
                       'sq': 'SmoothQuant per-channel weights, static per-tensor activations, int8 KV (configs[3], the benchmarked one)',
                       'sq_dyn': 'SmoothQuant per-channel weights, per-token dynamic activations, int8 KV (--per_token --per_channel)'}}
    # how decisive the reference's own choices are: a greedy token is only comparable where top-1 leads top-2 by more than
    # the logit error - a random-weight 32-layer model has very small margins (its logits barely depend on the prompt)
    # the noise floor of the free-running comparison: HF itself in fp16 (what the reference's own run_hf.py / summarize.py baseline
    # runs, Q/run_hf.py:55-57 `.half().cuda()`) against HF fp32 on the same weights - how far a mere change of precision moves a
    # greedy continuation of this parent
    try:
        with torch.no_grad():
            seq16, lg16 = run_hf.hf_generate(parent, prompts.to(dev), N, eos_token_id=None, pad_token_id=0, return_logits=True)
        t16 = seq16[:, P:].cpu().numpy()
        rl16 = [rouge_l_ids(t16[i], hf_tokens[i]) for i in range(n_prompts)]
        with torch.no_grad():
            tf16 = ref(seq16[:, :-1]).logits[:, P - 1:].float().cpu().numpy()
        e16 = np.abs(lg16.permute(1, 0, 2).cpu().numpy() - tf16)
        res['hf_fp16_vs_hf_fp32'] = {'what': 'HF LlamaForCausalLM in fp16 on the GPU (the reference\'s own baseline precision) against HF fp32, same weights',
                                     'rougeL_mean': float(np.mean(rl16)), 'rougeL_per_prompt': [round(x, 2) for x in rl16],
                                     'token_match_rate_free_running': float((t16 == hf_tokens).mean()),
                                     'max_abs_logit_err_teacher_forced': float(e16.max()), 'mean_abs_logit_err': float(e16.mean())}
    except Exception as e:  # side report
        res['hf_fp16_vs_hf_fp32'] = {'error': repr(e)}
    top2 = np.sort(hf_logits, axis=-1)[..., -2:]
    margin = top2[..., 1] - top2[..., 0]
    res['hf_top1_top2_margin'] = {'min': float(margin.min()), 'median': float(np.median(margin)), 'max': float(margin.max()),
                                  'fraction_below_0.1': float((margin < 0.1).mean())}
    cyc = [len(set(int(x) for x in hf_tokens[i, N // 2:])) for i in range(n_prompts)]
    res['hf_distinct_tokens_in_second_half'] = cyc  # a small number = the synthetic parent has fallen into a short cycle

    def teacher_forced(model_fwd, toks):
        """logits of `model_fwd` on every engine step's prefix: [prompts, new, vocab]"""
        full = torch.cat([prompts, torch.from_numpy(toks[:, :-1])], dim=1).to(dev)
        with torch.no_grad():
            return model_fwd(full).float().cpu().numpy()

    hf_fwd = lambda full: ref(full).logits[:, P - 1:]
    tf_ref = {}
    for mode in MODES:
        gl, gt = gpu[mode]['logits'], gpu[mode]['tokens']
        tf = teacher_forced(hf_fwd, gt)
        tf_ref[mode] = tf
        err = np.abs(gl - tf)
        rl = [rouge_l_ids(gt[i], hf_tokens[i]) for i in range(n_prompts)]
        div = [int(np.nonzero(gt[i] != hf_tokens[i])[0][0]) if (gt[i] != hf_tokens[i]).any() else None for i in range(n_prompts)]
        agree = gl.argmax(-1) == tf.argmax(-1)
        # agreement where HF itself is decisive (its top-1 / top-2 margin on that prefix exceeds twice the engine's error there)
        t2 = np.sort(tf, axis=-1)[..., -2:]
        decisive = (t2[..., 1] - t2[..., 0]) > 2 * err.max(-1)
        res[mode] = {
            'max_abs_logit_err': _err_stats(np, err, steps),
            'mean_abs_logit_err': float(err.mean()),
            'p99_abs_logit_err': float(np.quantile(err.max(-1), 0.99)),
            'within_reference_atol_1e-1': bool(err.max() < 1e-1),
            'argmax_agreement_on_same_prefix': float(agree.mean()),
            'argmax_agreement_where_hf_margin_exceeds_2x_error': {'steps': int(decisive.sum()),
                                                                  'agreement': float(agree[decisive].mean()) if decisive.any() else None},
            'token_match_rate_free_running': float((gt == hf_tokens).mean()),
            'first_divergent_step_per_prompt': div,
            'rougeL_vs_hf_per_prompt': [round(x, 2) for x in rl],
            'rougeL_vs_hf_mean': float(np.mean(rl)),
        }
    for mode in MODES[1:]:
        res[mode]['rougeL_delta_vs_fp16_engine'] = res['fp16']['rougeL_vs_hf_mean'] - res[mode]['rougeL_vs_hf_mean']
    base16 = res.get('hf_fp16_vs_hf_fp32', {}).get('rougeL_mean')
    if base16 is not None:  # the reference's own comparison: engine vs the HF fp16 baseline, both scored against the same target
        for mode in MODES:
            res[mode]['rougeL_delta_vs_hf_fp16_baseline'] = base16 - res[mode]['rougeL_vs_hf_mean']

    # ---- attribution of the SmoothQuant engine's distance to HF: kernels or algorithm?
    t1 = time.perf_counter()
    try:
        fq = FakeQuantSQ(torch, sq_tensors, layers)
        fq_logits = teacher_forced(lambda full: fq.forward(full, P, first_row=P - 1), gpu['sq']['tokens'])
        fq32 = FakeQuantSQ(torch, sq_tensors, layers, reduce_dtype=torch.float32)
        fq32_logits = teacher_forced(lambda full: fq32.forward(full, P, first_row=P - 1), gpu['sq']['tokens'])
        # the restatement with the ENGINE's rounding points in the attention (as_built): what is left between it and the engine
        # is summation order - to be compared with the control
        fqb = FakeQuantSQ(torch, sq_tensors, layers, as_built=True)
        fqb_logits = teacher_forced(lambda full: fqb.forward(full, P, first_row=P - 1), gpu['sq']['tokens'])
        e_built = np.abs(gpu['sq']['logits'] - fqb_logits)
        e_points = np.abs(fqb_logits - fq_logits)
        e_kernel = np.abs(gpu['sq']['logits'] - fq_logits)
        e_algo = np.abs(fq_logits - tf_ref['sq'])
        e_ctrl = np.abs(fq32_logits - fq_logits)
        e_kernel32 = np.abs(gpu['sq']['logits'] - fq32_logits)
        res['sq_attribution'] = {
            'what': 'torch restatement of the SmoothQuant-static + int8-KV ALGORITHM on the engine\'s own int8 weights and scales '
                    '(exact integer GEMMs, reference rounding points, fp64 reductions), teacher-forced on the engine\'s tokens',
            'engine_vs_algorithm_max_abs_logit_err': _err_stats(np, e_kernel, steps),
            'engine_vs_algorithm_mean_abs_logit_err': float(e_kernel.mean()),
            'algorithm_vs_hf_max_abs_logit_err': _err_stats(np, e_algo, steps),
            'algorithm_vs_hf_mean_abs_logit_err': float(e_algo.mean()),
            'engine_vs_hf_max_abs_logit_err': res['sq']['max_abs_logit_err']['all_steps'],
            'engine_vs_hf_mean_abs_logit_err': res['sq']['mean_abs_logit_err'],
            # the control: two restatements of the SAME algorithm on the same integers, fp64 vs fp32 reductions
            'control_algorithm_fp64_vs_fp32_reductions_max_abs_logit_err': _err_stats(np, e_ctrl, steps),
            'control_algorithm_fp64_vs_fp32_reductions_mean_abs_logit_err': float(e_ctrl.mean()),
            # r04: the same with the engine's two later rounding points (int8 KV used as scaled integers, un-normalised fp16
            # probabilities) in the restatement; and how far those two rounding points alone move the restatement
            'engine_vs_algorithm_as_built_max_abs_logit_err': _err_stats(np, e_built, steps),
            'engine_vs_algorithm_as_built_mean_abs_logit_err': float(e_built.mean()),
            'algorithm_reference_rounding_vs_as_built_max_abs_logit_err': float(e_points.max()),
            'algorithm_reference_rounding_vs_as_built_mean_abs_logit_err': float(e_points.mean()),
            'engine_vs_algorithm_fp32_reductions_max_abs_logit_err': float(e_kernel32.max()),
            'engine_vs_algorithm_fp32_reductions_mean_abs_logit_err': float(e_kernel32.mean()),
            'argmax_agreement_engine_vs_algorithm': float((gpu['sq']['logits'].argmax(-1) == fq_logits.argmax(-1)).mean()),
            'argmax_agreement_control': float((fq32_logits.argmax(-1) == fq_logits.argmax(-1)).mean()),
            'reading': 'engine_vs_hf ~ algorithm_vs_hf (same mean, same maximum) and engine_vs_algorithm ~ control: the engine is as '
                       'far from a restatement of its algorithm as two restatements that differ only in summation order are from '
                       'each other (one int8 flip ahead of a quantiser is amplified by the following quantised layers); the '
                       'distance to HF is the quantisation algorithm\'s',
        }
        log(f'SmoothQuant attribution (torch fake-quant, {n_prompts} teacher-forced forwards): {time.perf_counter() - t1:.1f} s')
    except Exception as e:  # side report
        import traceback
        traceback.print_exc(file=sys.stderr)
        res['sq_attribution'] = {'error': repr(e)}
    del sq_tensors
    torch.cuda.empty_cache()

    # ---- BASELINE.json configs[0]: HF fp32 on the host CPU (timed; cross-check of the GPU generator on prompt 0)
    cpu = cpu_info = None
    if cpu_leg:
        t1 = time.perf_counter()
        cpu = to_cpu_fp32(torch, parent, hf_cfg)
        torch.set_num_threads(cpu_threads)
        build_s = time.perf_counter() - t1
        t1 = time.perf_counter()
        seq_c, cpu_logits = run_hf.hf_generate(cpu, prompts[:1], cpu_new_tokens, eos_token_id=None, pad_token_id=0, return_logits=True)
        latency = time.perf_counter() - t1
        cpu_tokens = seq_c[0, P:].numpy()
        cpu_logits = cpu_logits[:, 0].numpy()
        same = bool(np.array_equal(cpu_tokens, hf_tokens[0, :cpu_new_tokens]))
        n_same = int(np.argmin(np.concatenate([cpu_tokens == hf_tokens[0, :cpu_new_tokens], [False]])))  # common prefix
        res['hf_gpu_vs_hf_cpu'] = {'prompt': 0, 'new_tokens': cpu_new_tokens, 'tokens_identical': same,
                                   'max_abs_logit_diff_on_common_prefix': float(np.abs(cpu_logits[:max(n_same, 1)]
                                                                                       - hf_logits[0, :max(n_same, 1)]).max())}
        log(f'HF-CPU: model build {build_s:.1f} s, generate({P} + {cpu_new_tokens}) {latency:.1f} s')
        cpu_info = dict(build_s=build_s, latency_s=latency, prompt_len=P, new_tokens=cpu_new_tokens,
                        tokens_per_s=cpu_new_tokens / latency, threads=cpu_threads)
    del ref, parent, sd, act
    torch.cuda.empty_cache()
    return res, cpu, cpu_info


# ----------------------------------------------------------------------------------------------------------------------------
# The same question on a TRAINED parent (tests/golden/trained_llama, made by tests/golden/train_tiny_llama.py): "ROUGE-L delta vs
# HF <= 1" cannot be decided by free-running generation on random weights (margins below the int8 noise); on
# a parent whose greedy continuations mean something it can.  The flow is the product's own command line, as the reference runs
# it (hf_llama_convert.py -> build.py -> summarize.py --test_hf --test_trt_llm, Q/summarize.py:91,260,321-323,352): 24 prompts x
# 100 new tokens, HF fp32 beside the engine, ROUGE of both against the language's own continuation (the `highlights`).
# tests/test_gpu_trained_accuracy.py asserts the same numbers per configuration.
# ----------------------------------------------------------------------------------------------------------------------------
TRAINED_CONFIGS = {
    'fp16': (False, []),
    'int8_kv': (False, ['--int8_kv_cache']),
    'woq8_int8kv': (False, ['--use_weight_only', '--int8_kv_cache']),
    'woq4_int8kv': (False, ['--use_weight_only', '--weight_only_precision', 'int4', '--int8_kv_cache']),
    'sq_static_int8kv': (True, ['--use_smooth_quant', '--per_channel', '--int8_kv_cache']),
    'sq_per_token_int8kv': (True, ['--use_smooth_quant', '--per_token', '--per_channel', '--int8_kv_cache']),
}


def trained_parent_report(log=print, configs=None, workdir=None):
    import json
    import subprocess
    import tempfile

    import numpy as np
    fix = os.path.join(ROOT, 'tests', 'golden', 'trained_llama')
    if not os.path.exists(os.path.join(fix, 'eval.npz')):
        return {'error': 'tests/golden/trained_llama is missing'}
    base = workdir or tempfile.mkdtemp(prefix='tllm_trained_')
    e = np.load(os.path.join(fix, 'eval.npz'))
    for k in ('prompts', 'lengths', 'reference', 'calib'):
        np.save(os.path.join(base, k + '.npy'), e[k])
    new = int(e['hf_tokens'].shape[1])
    info = json.load(open(os.path.join(fix, 'TRAINLOG.json')))
    out = {'parent': 'tests/golden/trained_llama (D 256, 4 layers, 4 heads x 64, FFN 768, vocab 512; trained by '
                     'tests/golden/train_tiny_llama.py)', 'prompts': int(e['prompts'].shape[0]), 'new_tokens': new,
           'hf_margin_median': info['margin']['median'], 'hf_margin_frac_below_0p2': info['margin']['frac_below_0p2'],
           'criterion': '|ROUGE-L(engine vs highlights) - ROUGE-L(HF fp32 vs highlights)| <= 1 (README.md:921, summarize.py)',
           'configs': {}}
    ft = {}
    py = sys.executable
    for name, (sq, flags) in TRAINED_CONFIGS.items():
        if configs and name not in configs:
            continue
        t0 = time.time()
        try:
            if sq not in ft:
                d = os.path.join(base, 'ft_sq' if sq else 'ft')
                subprocess.run([py, os.path.join(EX, 'hf_llama_convert.py'), '-i', fix, '-o', d, '--calibrate-kv-cache', '--calib-ids',
                                os.path.join(base, 'calib.npy')] + (['-sq', '0.5'] if sq else []), check=True, cwd=EX, timeout=900,
                               capture_output=True)
                ft[sq] = os.path.join(d, '1-gpu')
            eng = os.path.join(base, 'eng_' + name)
            subprocess.run([py, os.path.join(EX, 'build.py'), '--model_dir', ft[sq], '--output_dir', eng, '--max_batch_size', '4',
                            '--max_input_len', '256', '--max_output_len', str(new), '--log_level', 'error'] + flags, check=True, cwd=EX,
                           timeout=900, capture_output=True)
            res = os.path.join(base, f'rouge_{name}.json')
            subprocess.run([py, os.path.join(EX, 'summarize.py'), '--hf_model_location', fix, '--test_hf', '--test_trt_llm', '--data_type',
                            'fp32', '--engine_dir', eng, '--prompts_npy', os.path.join(base, 'prompts.npy'), '--prompt_lengths_npy',
                            os.path.join(base, 'lengths.npy'), '--references_npy', os.path.join(base, 'reference.npy'), '--output_len',
                            str(new), '--batch_size', '4', '--max_ite', str(int(e['prompts'].shape[0]) // 4), '--log_level', 'error',
                            '--output_json', res], check=True, cwd=EX, timeout=1800, capture_output=True)
            r = json.load(open(res))
            out['configs'][name] = {'rougeL': r['tensorrt_llm']['rougeL'], 'hf_rougeL': r['hf']['rougeL'],
                                    'rougeL_delta_vs_hf': r['rougeL_delta_vs_hf'], 'within_1': abs(r['rougeL_delta_vs_hf']) <= 1.0,
                                    'rougeL_of_engine_text_vs_hf_text': r['tensorrt_llm_vs_hf']['rougeL'],
                                    'token_match_rate': r['token_match_rate'], 'seconds': time.time() - t0}
            log(f'trained parent, {name}: ROUGE-L {r["tensorrt_llm"]["rougeL"]:.2f} (HF {r["hf"]["rougeL"]:.2f}, delta '
                f'{r["rougeL_delta_vs_hf"]:+.2f}), token match {r["token_match_rate"]:.3f}')
        except Exception as ex:  # side report: one failing configuration must not cost the others
            msg = getattr(ex, 'stderr', b'')
            out['configs'][name] = {'error': repr(ex), 'stderr_tail': (msg.decode(errors='replace')[-600:] if msg else '')}
    return out
