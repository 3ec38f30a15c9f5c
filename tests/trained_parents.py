"""Shared by the accuracy tests on the two trained parents (tests/golden/trained_llama: a deterministic language, HF margins
>= 5.5; tests/golden/trained_llama_stochastic: 2 - 4 near-equiprobable continuations at ~30 % of the positions, 29 % of HF's
steps with a margin below 1.0).  Not a test module.

THE LOGIT BOUND OF A QUANTISED ENGINE, as a principle instead of a fitted number ("self-fitted LOGIT_TOL"):
the reference states a logit tolerance for an fp16 model only (atol 1e-1, T/tests/model/test_llama.py:286-288, 352-354); a
quantised ALGORITHM - int8 KV cache, weight-only int8 / int4, SmoothQuant - has an error of its own against HF fp32 that no
implementation can undercut.  oracle/quant_oracle.py restates each algorithm in numpy with the reference's rounding points, on
the same integers and scales the engine gets.  On HF's own token path (teacher forced):

    max |logit(engine) - logit(HF)|   <=  K * max |logit(oracle) - logit(HF)|  +  A
    mean |logit(engine) - logit(HF)|  <=  K * mean |logit(oracle) - logit(HF)| +  A / 10

with K = 1.25 and A = 1e-1 (the reference's own fp16 tolerance: what two correct fp16 implementations may differ by), stated
ONCE here for every configuration and both parents.  A kernel regression cannot hide inside the bound: it would have to stay
below a quarter of the algorithm's own error."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K_ALGORITHM = 1.25  # engine error <= K x the restated algorithm's error (+ the fp16 allowance A)
A_FP16 = 1e-1       # T/tests/model/test_llama.py:288,354

PARENTS = {'deterministic': os.path.join(ROOT, 'tests', 'golden', 'trained_llama'),
           'stochastic': os.path.join(ROOT, 'tests', 'golden', 'trained_llama_stochastic')}

# oracle mode -> int8 KV cache
ORACLE_MODES = {'fp16': 0, 'int8_kv': 1, 'woq8_int8kv': 1, 'woq4_int8kv': 1, 'sq_static_int8kv': 1, 'sq_per_token_int8kv': 1}
_QO_MODE = {'fp16': 'fp16', 'int8_kv': 'fp16', 'woq8_int8kv': 'woq8', 'woq4_int8kv': 'woq4', 'sq_static_int8kv': 'sq_static_pc',
            'sq_per_token_int8kv': 'sq_dyn_pc'}


def load_eval(parent):
    e = np.load(os.path.join(PARENTS[parent], 'eval.npz'))
    return {k: e[k] for k in e.files}


def load_config(parent):
    return json.load(open(os.path.join(PARENTS[parent], 'config.json')))


def oracle_weights(parent):
    """(cfg, w) in the oracle's / engine's naming from the HF checkpoint of the fixture: qkv = [q; k; v] rows, mlp.fc = gate_proj
    (the SiLU branch), mlp.gate = up_proj, mlp.proj = down_proj (Q/weight.py: load_from_hf_llama)."""
    from safetensors.numpy import load_file
    hf = load_config(parent)
    t = load_file(os.path.join(PARENTS[parent], 'model.safetensors'))
    L = hf['num_hidden_layers']
    w = {'vocab_embedding.weight': t['model.embed_tokens.weight'], 'ln_f.weight': t['model.norm.weight'], 'lm_head.weight': t['lm_head.weight']}
    for i in range(L):
        p, q = f'layers.{i}.', f'model.layers.{i}.'
        w[p + 'input_layernorm.weight'] = t[q + 'input_layernorm.weight']
        w[p + 'post_layernorm.weight'] = t[q + 'post_attention_layernorm.weight']
        w[p + 'attention.qkv.weight'] = np.concatenate([t[q + 'self_attn.q_proj.weight'], t[q + 'self_attn.k_proj.weight'],
                                                        t[q + 'self_attn.v_proj.weight']], 0)
        w[p + 'attention.dense.weight'] = t[q + 'self_attn.o_proj.weight']
        w[p + 'mlp.fc.weight'] = t[q + 'mlp.gate_proj.weight']
        w[p + 'mlp.gate.weight'] = t[q + 'mlp.up_proj.weight']
        w[p + 'mlp.proj.weight'] = t[q + 'mlp.down_proj.weight']
    w = {k: np.ascontiguousarray(v.astype(np.float16)) for k, v in w.items()}
    cfg = dict(num_layers=L, num_heads=hf['num_attention_heads'], hidden_size=hf['hidden_size'], inter_size=hf['intermediate_size'],
               vocab_size=hf['vocab_size'], max_position_embeddings=hf['max_position_embeddings'], rms_norm_eps=hf['rms_norm_eps'])
    return cfg, w


def quantised(parent, name):
    """oracle/quant_oracle.quantise_model of the parent for configuration `name`, calibrated on the fixture's calibration prompts."""
    from oracle import quant_oracle as QO
    cfg, w = oracle_weights(parent)
    e = load_eval(parent)
    calib = e['calib'][:16].astype(np.int32)
    return cfg, QO.quantise_model(cfg, w, _QO_MODE[name], ORACLE_MODES[name], calib_ids=calib,
                                  calib_lens=np.full(calib.shape[0], calib.shape[1], np.int32))


def teacher_forced_prompts(parent, n):
    """The first n prompts of the fixture whose per-step HF logits it holds, right-padded to a common length, with HF's tokens."""
    e = load_eval(parent)
    n = min(n, e['hf_logits'].shape[0])
    lens = e['lengths'][:n].astype(np.int32)
    S = int(lens.max())
    ids = np.zeros((n, S), np.int32)
    for b in range(n):
        ids[b, :lens[b]] = e['prompts'][b, :lens[b]]
    return ids, lens, e['hf_tokens'][:n].astype(np.int32), e['hf_logits'][:n].astype(np.float32), float(e['hf_logits_absmax'])


def oracle_logits(qmodel, ids, lens, hf_tokens, steps):
    """[n, steps, V]: the restated algorithm on HF's token path (logits of generation step s = the distribution of token s)."""
    from oracle import quant_oracle as QO
    ref, _ = QO.run_model(qmodel, ids, lens, steps, feed_ids=hf_tokens[:, :steps])
    return np.stack([np.asarray(r, np.float32) for r in ref[:steps]], 1)


def engine_logits(qmodel, cfg, ids, lens, hf_tokens, steps):
    from tensorrt_llm.runtime.native import NativeSession
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    n, S = ids.shape
    s.setup(n, S, steps)
    s.context(ids, lens)
    out = np.zeros((n, steps, cfg['vocab_size']), np.float32)
    for st in range(steps):
        out[:, st] = s.logits()
        if st + 1 < steps:
            s.force_tokens(hf_tokens[:, st])
            s.step(1, use_graph=False)
    s.close()
    return out
