"""HF LLaMA state dict -> the tensors of a tp = 1 engine, in memory.

The same conversion as `hf_llama_convert.py` -> FT directory -> `weight.py::load_from_ft_llama`, without the 30 GB of files in
between (a 7B model): calibration (`smoothquant.capture_activation_range`), SmoothQuant (`hf_llama_convert.smooth_llama_model`),
int8 generation (`convert.generate_int8`) and the weight-only quantiser (`tllm_symmetric_quantize_last_axis`) are the shipped
ones; this file only does the naming / transposition `load_from_ft_llama` does (reference: Q/weight_quant.py:116-147, :227-446).
tests/test_convert.py holds the two routes to the same bytes.  Tensors stay on the device the state dict lives on, so the
result can be handed to `tllm_session_set_tensor(location = 1)` directly (bench.py's parity run at LLaMA-7B)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from convert import generate_int8
from hf_llama_convert import smooth_llama_model

HF_LINEARS = (('attention.dense', 'self_attn.o_proj'), ('mlp.fc', 'mlp.gate_proj'), ('mlp.gate', 'mlp.up_proj'),
              ('mlp.proj', 'mlp.down_proj'))


def _f32(x, dev):
    return torch.from_numpy(np.array(x, dtype=np.float32, copy=True)).to(dev)


@torch.no_grad()
def engine_tensors(sd, num_layers, mode='fp16', act_range=None, alpha=0.5, per_channel=True, per_token=False,
                   int8_kv=False, num_heads=None, threads=16, alpha_down=None):
    """sd: HF-named state dict (torch tensors, one device).  mode: 'fp16' | 'woq8' | 'woq4' | 'sq'.
    `act_range` (capture_activation_range of the UN-smoothed model) is needed for 'sq' and for int8_kv; with 'sq' the
    state dict is smoothed on a float32 copy first.  Returns {engine tensor name: torch tensor}."""
    dev = sd['model.embed_tokens.weight'].device
    out = {'vocab_embedding.weight': sd['model.embed_tokens.weight'].half().contiguous(),
           'ln_f.weight': sd['model.norm.weight'].half().contiguous(),
           'lm_head.weight': sd.get('lm_head.weight', sd['model.embed_tokens.weight']).half().contiguous()}  # fp16 in every mode
    if mode == 'sq':
        assert act_range is not None and num_heads is not None
        act_range = {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in act_range.items()}
        sd = {k: (v.detach().float().clone() if '.layers.' in k else v) for k, v in sd.items()}
        smooth_llama_model(sd, act_range, alpha, num_layers, num_heads, num_heads, alpha_down=alpha_down)
    woq_jobs = []
    for i in range(num_layers):
        hp, p = f'model.layers.{i}.', f'layers.{i}.'
        out[p + 'input_layernorm.weight'] = sd[hp + 'input_layernorm.weight'].half().contiguous()
        out[p + 'post_layernorm.weight'] = sd[hp + 'post_attention_layernorm.weight'].half().contiguous()
        q, k, v = (sd[hp + f'self_attn.{n}_proj.weight'] for n in 'qkv')
        mats = [('attention.qkv', torch.cat([q, k, v], dim=0))] + [(n, sd[hp + hf + '.weight']) for n, hf in HF_LINEARS]
        qkv_range = None
        if act_range is not None:
            r = [act_range[hp + f'self_attn.{n}_proj'] for n in 'qkv']
            qkv_range = {'x': r[0]['x'], 'y': torch.cat([t['y'] for t in r]), 'w': torch.cat([t['w'] for t in r])}
        if mode == 'fp16':
            for n, w in mats:
                out[p + n + '.weight'] = w.half().contiguous()
        elif mode in ('woq8', 'woq4'):
            for n, w in mats:
                woq_jobs.append((p + n, w))
        elif mode == 'sq':
            suffix = '.col' if per_channel else ''
            key = 'scale_w_quant_orig' if per_token else 'scale_y_accum_quant'
            scale_x = {}
            for n, w in mats:
                is_qkv = n == 'attention.qkv'
                rng = qkv_range if is_qkv else act_range[hp + dict(HF_LINEARS)[n]]
                D_out = w.shape[0]
                # FT orientation: [in, out], QKV [in, 3, out]  (hf_llama_convert.py)
                w_ft = w.t().reshape(w.shape[1], 3, D_out // 3) if is_qkv else w.t()
                vals = generate_int8(w_ft, rng, is_qkv=is_qkv)
                w8 = vals['weight.int8' + suffix]
                # back to [out, in]  (load_from_ft_llama::set_sq)
                out[p + n + '.weight'] = (w8.permute(1, 2, 0).reshape(D_out, -1) if is_qkv else w8.t()).contiguous()
                s = np.asarray(vals[key + suffix], np.float32)
                if is_qkv:
                    s = s.reshape(3, -1)  # per channel, or one factor each for Q, K, V broadcast to [3, out]: a vector either way
                    s = np.broadcast_to(s, (3, D_out // 3)).reshape(1, -1)
                elif per_channel:
                    s = s.reshape(1, -1)
                else:
                    s = s.reshape(-1)[:1].reshape(1, 1)
                out[p + n + '.per_channel_scale'] = _f32(s, dev)
                if not per_token:
                    out[p + n + '.act_scale'] = _f32(np.asarray(vals['scale_y_quant_orig']).reshape(-1)[:1].reshape(1, 1), dev)
                scale_x[n] = _f32(np.asarray(vals['scale_x_orig_quant']).reshape(-1)[:1], dev)
                if is_qkv:
                    kv_scale = np.asarray(vals['scale_y_quant_orig'], np.float32).reshape(-1)[:1]
            if not per_token:
                out[p + 'input_layernorm.scale_to_int'] = scale_x['attention.qkv']
                out[p + 'attention.quantization_scaling_factor'] = scale_x['attention.dense']
                out[p + 'post_layernorm.scale_to_int'] = scale_x['mlp.fc']
                out[p + 'mlp.quantization_scaling_factor'] = scale_x['mlp.proj']
        else:
            raise ValueError(mode)
        if int8_kv:
            # kv_quant_orig = scale_y_quant_orig of the QKV output, kv_orig_quant = 1 / that (Q/weight_quant.py:439-446)
            if mode != 'sq':
                assert qkv_range is not None, 'int8 KV cache needs calibrated activation ranges'
                kv_scale = np.array([float(qkv_range['y'].max()) / 127.0], np.float32)
            out[p + 'attention.kv_quant_orig_scale'] = _f32(kv_scale, dev)
            out[p + 'attention.kv_orig_quant_scale'] = _f32(1.0 / kv_scale, dev)
    if woq_jobs:
        from tensorrt_llm.plugin import capi
        bits = 8 if mode == 'woq8' else 4

        def job(item):
            name, w = item
            # the shipped host quantiser takes the reference's [k, n] orientation (thop/weightOnlyQuantOp.cpp:143-236)
            processed, scales, _ = capi.symmetric_quantize_last_axis(w.detach().half().t().contiguous().cpu().numpy(), bits)
            return name, processed, scales

        with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:  # ctypes releases the GIL inside the C call
            for name, processed, scales in pool.map(job, woq_jobs):
                out[name + '.weight'] = torch.from_numpy(processed).to(dev)
                out[name + '.per_channel_scale'] = torch.from_numpy(scales).to(dev)
    return out
