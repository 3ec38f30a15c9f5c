"""The accuracy half of BASELINE.json's metric ("... ; ROUGE-L delta vs HF", logits within a stated fp16 tolerance) at LLaMA-7B,
and BASELINE.json configs[0] (HF transformers on the host CPU, batch 1, prompt 128) - called by bench.py, which puts the
result into its JSON line as `parity` (and reuses the CPU model for `cpu_baseline`).

No checkpoint, dataset or tokenizer exists on the GPU box, so this is the substitute SURVEY.md section 8d specifies:

  * ONE seeded fp16 parent model of the 7B architecture (Xavier-uniform matrices as tensorrt_llm.Parameter draws them,
    PY/parameter.py:28-38; RMSNorm weights 1 + 0.1 U(-1, 1); embeddings N(0, 0.02); 1 % of the hidden channels carry
    x20 outliers so that SmoothQuant has something to smooth), built ON THE GPU as an HF LlamaForCausalLM;
  * the same weights, bit for bit, copied down into an HF fp32 model on the host CPU = the reference's own accuracy oracle
    (Q/run_hf.py:41-104, Q/summarize.py:207-216, T/tests/model/test_llama.py:286-288 - `run_hf.hf_generate`);
  * the product engines from that parent through the product's own converter (examples/llama_quant/inmemory.py ->
    smoothquant.capture_activation_range, hf_llama_convert.smooth_llama_model, convert.generate_int8,
    tllm_symmetric_quantize_last_axis): fp16, weight-only int8 + int8 KV, SmoothQuant per-channel static int8 + int8 KV
    (BASELINE.json configs[1..3]); calibration = seeded random prompts through the HF model on the GPU, as the reference's
    hf_llama_convert.py runs it (torch there is the calibration tool, not the product path);
  * shape = BASELINE.json configs[0]: batch 1, prompt 128, 16 new tokens, greedy, EOS stopping off.

Per configuration: max |logit error| at steps 0 / 1 / last and over all steps - HF-CPU evaluated ON THE GPU RUN'S OWN TOKEN
PATH (one teacher-forced forward), so a flipped near-tie does not turn every later step into a comparison of two different
sentences; arg-max agreement on that path; free-running token match and in-repo ROUGE-L (summarize.py) of the generated
token strings against HF-CPU's free-running greedy output, and its delta to the fp16 engine's score."""
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


def to_cpu_fp32(torch, parent, cfg):
    """HF fp32 model on the host CPU holding exactly the parent's (fp16-representable) weights."""
    import transformers
    with torch.device('meta'):
        m = transformers.LlamaForCausalLM(cfg)
    m = m.to_empty(device='cpu').float().eval()
    src = dict(parent.named_parameters())
    with torch.no_grad():
        for name, p in m.named_parameters():
            p.copy_(src[name].detach().float().cpu())
        if hasattr(m.model, 'rotary_emb'):
            m.model.rotary_emb = type(m.model.rotary_emb)(config=cfg)
    return m


def rouge_l_ids(pred_ids, ref_ids):
    """ROUGE-L F-measure x 100 of two token-id sequences through the product's own implementation (summarize.py)."""
    import summarize
    return 100.0 * summarize.rouge_l(' '.join(str(int(t)) for t in pred_ids), ' '.join(str(int(t)) for t in ref_ids))


QM = dict(fp16=0, woq8=2, sq=2 | 4 | 8, sq_dyn=2 | 4 | 8 | 16)
MODES = ('fp16', 'woq8', 'sq', 'sq_dyn')
INT8_KV = 32


def run(torch, dev, layers=32, prompt_len=128, new_tokens=16, cpu_threads=64, calib_samples=16, calib_len=128, log=None):
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
    prompt = torch.randint(3, 32000, (1, prompt_len), generator=g)  # ids 0-2 reserved, seed 1
    gc = torch.Generator().manual_seed(2)
    calib = [torch.randint(3, 32000, (1, calib_len), generator=gc) for _ in range(calib_samples)]
    act = smoothquant.capture_activation_range(parent, calib, num_samples=calib_samples)
    log(f'parent + calibration: {time.perf_counter() - t0:.1f} s')
    sd = dict(parent.state_dict())
    cfg = dict(num_layers=layers, num_heads=32, hidden_size=4096, inter_size=11008, vocab_size=32000, max_position_embeddings=2048,
               rms_norm_eps=1e-6)
    ids_np = prompt.numpy().astype(np.int32)
    lens = np.array([prompt_len], np.int32)
    gpu = {}
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
        s.setup(1, prompt_len, new_tokens)
        stream = torch.cuda.current_stream().cuda_stream
        s.context(ids_np, lens, stream=stream)
        logits = [s.logits(stream=stream)[0]]
        for k in range(1, new_tokens):
            s.step(1, use_graph=k > 1, stream=stream)  # first step eager, the rest replayed from the step's hipGraph
            logits.append(s.logits(stream=stream)[0])
        toks = s.output_ids(stream=stream)[0, prompt_len:prompt_len + new_tokens].copy()
        s.close()
        del tensors
        torch.cuda.empty_cache()
        gpu[mode] = dict(logits=np.stack(logits), tokens=toks)
        log(f'{mode}: converted + generated in {time.perf_counter() - t1:.1f} s')
    # ---- the reference path: HF fp32 on the host CPU, same weights
    t1 = time.perf_counter()
    cpu = to_cpu_fp32(torch, parent, hf_cfg)
    del parent, sd, act
    torch.cuda.empty_cache()
    torch.set_num_threads(cpu_threads)
    build_s = time.perf_counter() - t1
    t1 = time.perf_counter()
    seq, cpu_logits = run_hf.hf_generate(cpu, prompt, new_tokens, eos_token_id=None, pad_token_id=0, return_logits=True)
    latency = time.perf_counter() - t1
    cpu_tokens = seq[0, prompt_len:].numpy()
    cpu_logits = cpu_logits[:, 0].numpy()  # [new, vocab]
    scale = float(np.abs(cpu_logits).max())
    log(f'HF-CPU: model build {build_s:.1f} s, generate({prompt_len} + {new_tokens}) {latency:.1f} s')
    res = {'shape': f'batch 1, prompt {prompt_len}, {new_tokens} new tokens, greedy, EOS off (BASELINE.json configs[0] shape)',
           'weights': 'one seeded fp16 LLaMA-7B parent (Xavier, x20 outlier channels); HF fp32 on the host CPU holds the same values',
           'reference': f'HF transformers LlamaForCausalLM fp32 on {cpu_threads} CPU threads via run_hf.hf_generate',
           'layers': layers, 'logit_scale_max_abs': scale,
           'tolerance': 'reference bound: logits atol 1e-1 (T/tests/model/test_llama.py:286-288); ROUGE-L delta <= 1 (README.md:921)',
           'hf_cpu_tokens': [int(t) for t in cpu_tokens],
           'configs': {'fp16': 'fp16 + fp16 KV (BASELINE configs[1])', 'woq8': 'weight-only int8 + int8 KV (configs[2])',
                       'sq': 'SmoothQuant per-channel weights, static per-tensor activations, int8 KV (configs[3], the benchmarked one)',
                       'sq_dyn': 'SmoothQuant per-channel weights, per-token dynamic activations, int8 KV (--per_token --per_channel)'}}
    # how decisive the reference's own choices are: a greedy token is only comparable where top-1 leads top-2 by more than
    # the logit error - a random-weight 32-layer model has very small margins (its logits barely depend on the prompt)
    top2 = np.sort(cpu_logits, axis=-1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    res['hf_cpu_top1_top2_margin'] = {'min': float(margin.min()), 'median': float(np.median(margin)), 'max': float(margin.max())}
    for mode in MODES:
        gl, gt = gpu[mode]['logits'], gpu[mode]['tokens']
        if np.array_equal(gt, cpu_tokens):
            ref = cpu_logits  # same path: the free-running logits ARE the teacher-forced ones
        else:
            with torch.no_grad():
                full = torch.cat([prompt, torch.from_numpy(gt[:-1].astype(np.int64))[None]], dim=1)
                ref = cpu(full).logits[0, prompt_len - 1:].float().numpy()
        err = np.abs(gl - ref)
        div = np.nonzero(gt != cpu_tokens)[0]
        res[mode] = {
            'max_abs_logit_err': {'step_0': float(err[0].max()), 'step_1': float(err[1].max()),
                                  f'step_{new_tokens - 1}': float(err[-1].max()), 'all_steps': float(err.max())},
            'mean_abs_logit_err': float(err.mean()),
            'within_reference_atol_1e-1': bool(err.max() < 1e-1),
            'argmax_agreement_on_same_prefix': float(np.mean(gl.argmax(-1) == ref.argmax(-1))),
            'token_match_rate_free_running': float(np.mean(gt == cpu_tokens)),
            'first_divergent_step': int(div[0]) if len(div) else None,
            'rougeL_vs_hf_cpu': rouge_l_ids(gt, cpu_tokens),
            'tokens': [int(t) for t in gt],
        }
    for mode in MODES[1:]:
        res[mode]['rougeL_delta_vs_fp16_engine'] = res['fp16']['rougeL_vs_hf_cpu'] - res[mode]['rougeL_vs_hf_cpu']
    cpu_info = dict(build_s=build_s, latency_s=latency, prompt_len=prompt_len, new_tokens=new_tokens,
                    tokens_per_s=new_tokens / latency, threads=cpu_threads)
    return res, cpu, cpu_info
