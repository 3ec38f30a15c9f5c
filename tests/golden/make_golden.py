"""Generates the golden fixtures under tests/golden/ — run in the BUILD container only (needs /root/reference
and HF transformers on CPU); the fixtures are data (inputs + expected outputs), the reference sources never travel.

    python tests/golden/make_golden.py

What is pinned:
  hf_tiny_llama.npz   HF transformers LlamaForCausalLM (the accuracy oracle of T/tests/model/test_llama.py:166-173:
                      hidden 64, 2 heads, inter 24, vocab 128, 2 layers) on CPU in fp32 with fp16-representable
                      weights: logits of the context step and of one generation step for a padded batch of 2,
                      per-layer hidden states, plus LlamaRMSNorm / LlamaMLP outputs (T/tests/test_layer.py:98-196).
  quant_mode.json     truth table of T/tensorrt_llm/quantization/mode.py (imported by path).
  generate_int8.npz   outputs of examples/llama_quant/convert.py::generate_int8 (imported by path, with a 2-line
                      stub for tensorrt_llm._utils.torch_to_numpy) on seeded QKV [in,3,out] and dense [in,out].
  smooth_gemm.npz     outputs of examples/llama_quant/smoothquant.py::smooth_gemm on seeded weights / act scales.
"""
import importlib.util
import itertools
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/tensorrt_llm_july-release-v1'


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def make_hf_tiny():
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama.modeling_llama import LlamaMLP, LlamaRMSNorm
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, num_attention_heads=2, num_key_value_heads=2, intermediate_size=24, vocab_size=128,
                      num_hidden_layers=2, max_position_embeddings=64, rms_norm_eps=1e-6, hidden_act='silu',
                      attention_bias=False, tie_word_embeddings=False)
    cfg._attn_implementation = 'eager'
    model = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(4.0)  # livelier logits than the 0.02-std init
            p.copy_(p.half().float())  # fp16-representable weights
        for l in model.model.layers:  # non-trivial norm weights
            l.input_layernorm.weight.copy_((1 + 0.1 * torch.randn(64)).half().float())
            l.post_attention_layernorm.weight.copy_((1 + 0.1 * torch.randn(64)).half().float())
        model.model.norm.weight.copy_((1 + 0.1 * torch.randn(64)).half().float())
    B, S = 2, 8
    lens = [8, 5]
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(3, 128, (B, S), generator=g)
    for b in range(B):
        ids[b, lens[b]:] = 2  # pad id (T/examples/llama_quant/run.py:25-26)
    mask = torch.zeros(B, S, dtype=torch.long)
    for b in range(B):
        mask[b, :lens[b]] = 1
    pos = torch.arange(S)[None, :].repeat(B, 1)
    with torch.no_grad():
        out = model(input_ids=ids, attention_mask=mask, position_ids=pos, use_cache=True, output_hidden_states=True)
        logits_ctx = torch.stack([out.logits[b, lens[b] - 1] for b in range(B)])
        next_ids = logits_ctx.argmax(-1)
        # generation step: the new token sits at slot S for every sequence (padded layout), RoPE position = len[b]
        mask2 = torch.cat([mask, torch.ones(B, 1, dtype=torch.long)], dim=1)
        pos2 = torch.tensor(lens)[:, None]
        out2 = model(input_ids=next_ids[:, None], attention_mask=mask2, position_ids=pos2,
                     past_key_values=out.past_key_values, use_cache=True, output_hidden_states=True)
        logits_dec = out2.logits[:, 0]
    sd = model.state_dict()
    d = {
        'ids': ids.numpy().astype(np.int32),
        'input_lengths': np.array(lens, np.int32),
        'logits_ctx': logits_ctx.numpy(),
        'next_ids': next_ids.numpy().astype(np.int32),
        'logits_dec': logits_dec.numpy(),
        'hidden_ctx': torch.stack(out.hidden_states).numpy(),  # [L+1, B, S, D] (last one is after the final norm? no: HF
        # returns inputs of each layer + output of the last layer BEFORE norm in <=4.x, AFTER norm in 5.x: see test)
        'hidden_dec': torch.stack(out2.hidden_states).numpy(),
        'vocab_embedding.weight': sd['model.embed_tokens.weight'].numpy().astype(np.float16),
        'ln_f.weight': sd['model.norm.weight'].numpy().astype(np.float16),
        'lm_head.weight': sd['lm_head.weight'].numpy().astype(np.float16),
    }
    for i in range(2):
        pre = f'model.layers.{i}.'
        q, k, v = (sd[pre + f'self_attn.{n}_proj.weight'] for n in 'qkv')
        d[f'layers.{i}.attention.qkv.weight'] = torch.cat([q, k, v], 0).numpy().astype(np.float16)
        d[f'layers.{i}.attention.dense.weight'] = sd[pre + 'self_attn.o_proj.weight'].numpy().astype(np.float16)
        d[f'layers.{i}.input_layernorm.weight'] = sd[pre + 'input_layernorm.weight'].numpy().astype(np.float16)
        d[f'layers.{i}.post_layernorm.weight'] = sd[pre + 'post_attention_layernorm.weight'].numpy().astype(np.float16)
        d[f'layers.{i}.mlp.fc.weight'] = sd[pre + 'mlp.gate_proj.weight'].numpy().astype(np.float16)  # fc <-> gate_proj
        d[f'layers.{i}.mlp.gate.weight'] = sd[pre + 'mlp.up_proj.weight'].numpy().astype(np.float16)  # gate <-> up_proj
        d[f'layers.{i}.mlp.proj.weight'] = sd[pre + 'mlp.down_proj.weight'].numpy().astype(np.float16)
    # layer-level pins (test_layer.py:98-135 RMSNorm fp32 atol 1e-6; :137-196 GatedMLP fp32 atol 1e-5)
    torch.manual_seed(2)
    x = torch.randn(2, 8, 64)
    norm = LlamaRMSNorm(64, eps=1e-6)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(64))
        d['rms_x'] = x.numpy()
        d['rms_w'] = norm.weight.numpy()
        d['rms_y'] = norm(x).numpy()
        mlp = LlamaMLP(cfg)
        d['mlp_x'] = x.numpy()
        d['mlp_fc'] = mlp.gate_proj.weight.numpy().copy()
        d['mlp_gate'] = mlp.up_proj.weight.numpy().copy()
        d['mlp_proj'] = mlp.down_proj.weight.numpy().copy()
        d['mlp_y'] = mlp(x).numpy()
    np.savez_compressed(os.path.join(HERE, 'hf_tiny_llama.npz'), **d)
    import transformers
    return {'transformers': transformers.__version__, 'torch': torch.__version__}


def make_quant_mode():
    mode = load_by_path('ref_mode', os.path.join(REF, 'tensorrt_llm/quantization/mode.py'))
    Q = mode.QuantMode
    preds = ['is_int8_weight_only', 'is_int4_weight_only', 'is_weight_only', 'has_act_and_weight_quant',
             'has_per_token_dynamic_scaling', 'has_act_static_scaling', 'has_per_channel_scaling',
             'has_int8_kv_cache', 'has_fp8_kv_cache', 'has_any_quant']
    table = {str(v): {p: bool(getattr(Q(v), p)()) for p in preds} for v in range(128)}
    desc = []
    for args in itertools.product([False, True], repeat=7):
        try:
            desc.append([list(args), int(Q.from_description(*args))])
        except ValueError:
            desc.append([list(args), None])
    extra = {
        'use_smooth_quant': {f'{pt},{pc}': int(Q.use_smooth_quant(pt, pc)) for pt in (False, True) for pc in (False, True)},
        'use_weight_only': {str(i4): int(Q.use_weight_only(i4)) for i4 in (False, True)},
        'sq_pc_int8kv': int(Q.use_smooth_quant(False, True).set_int8_kv_cache()),
        'flags': {n: int(getattr(Q, n)) for n in ['INT4_WEIGHTS', 'INT8_WEIGHTS', 'ACTIVATIONS', 'PER_CHANNEL', 'PER_TOKEN',
                                                  'INT8_KV_CACHE', 'FP8_KV_CACHE', 'COUNT', 'WEIGHTS_AND_ACTIVATIONS',
                                                  'VALID_FLAGS']},
    }
    with open(os.path.join(HERE, 'quant_mode.json'), 'w') as f:
        json.dump({'predicates': table, 'from_description': desc, **extra}, f)


def make_generate_int8():
    stub = types.ModuleType('tensorrt_llm')
    stub_utils = types.ModuleType('tensorrt_llm._utils')
    stub_utils.torch_to_numpy = lambda x: x.detach().cpu().numpy()
    stub._utils = stub_utils
    sys.modules['tensorrt_llm'] = stub
    sys.modules['tensorrt_llm._utils'] = stub_utils
    conv = load_by_path('ref_convert', os.path.join(REF, 'examples/llama_quant/convert.py'))
    del sys.modules['tensorrt_llm'], sys.modules['tensorrt_llm._utils']
    torch.manual_seed(3)
    d = {}
    cin, cout = 32, 48
    # dense [in, out]
    w = torch.randn(cin, cout).numpy().astype(np.float32)
    rng = {'x': torch.rand(cin) * 4 + 0.1, 'y': torch.rand(cout) * 8 + 0.1, 'w': torch.from_numpy(np.abs(w).max(0))}
    res = conv.generate_int8(w, rng)
    d['dense_w'] = w
    for k in rng:
        d[f'dense_range_{k}'] = rng[k].numpy()
    for k, v in res.items():
        d[f'dense_out_{k}'] = np.asarray(v)
    # qkv [in, 3, out]
    wq = torch.randn(cin, 3, cout).numpy().astype(np.float32)
    rngq = {'x': torch.rand(cin) * 4 + 0.1, 'y': torch.rand(3 * cout) * 8 + 0.1,
            'w': torch.from_numpy(np.abs(wq).max(0).reshape(-1))}
    resq = conv.generate_int8(wq, rngq, is_qkv=True)
    d['qkv_w'] = wq
    for k in rngq:
        d[f'qkv_range_{k}'] = rngq[k].numpy()
    for k, v in resq.items():
        d[f'qkv_out_{k}'] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, 'generate_int8.npz'), **d)


def make_smooth_gemm():
    sq = load_by_path('ref_smoothquant', os.path.join(REF, 'examples/llama_quant/smoothquant.py'))
    torch.manual_seed(4)
    cin = 40
    w1, w2 = torch.randn(24, cin), torch.randn(56, cin)
    act = torch.rand(cin) * 5 + 0.01
    d = {'w1': w1.numpy().copy(), 'w2': w2.numpy().copy(), 'act': act.numpy().copy()}
    a, b = w1.clone(), w2.clone()
    s = sq.smooth_gemm([a, b], act, None, None, 0.5)
    d['s_joint'] = s.numpy()
    d['w1_joint'] = a.numpy()
    d['w2_joint'] = b.numpy()
    c = w1.clone()
    s1 = sq.smooth_gemm(c, act, None, None, 0.8)
    d['s_single_a08'] = s1.numpy()
    d['w1_single_a08'] = c.numpy()
    np.savez_compressed(os.path.join(HERE, 'smooth_gemm.npz'), **d)


if __name__ == '__main__':
    versions = make_hf_tiny()
    make_quant_mode()
    make_generate_int8()
    make_smooth_gemm()
    with open(os.path.join(HERE, 'VERSIONS.json'), 'w') as f:
        json.dump({**versions, 'numpy': np.__version__, 'reference': 'TRT2022/trtllm-llama @ v0 (/root/reference)'}, f)
    print('golden fixtures written to', HERE)
