"""SmoothQuant utilities for the LLaMA converter (T/examples/llama_quant/smoothquant.py).

Same entry points - capture_activation_range, smooth_gemm - with two corrections the reference needs to be usable on
LLaMA: (1) `w` statistics are per OUTPUT channel of an nn.Linear (the reference takes
`weight.abs().max(dim=0)` which is per input channel for Linear: it was written for GPT-2's Conv1D); (2) calibration
takes ready token ids, so it runs without a tokenizer or dataset download (offline boxes)."""
import functools
from collections import defaultdict

import torch
import torch.nn as nn


@torch.no_grad()
def smooth_gemm(gemm_weights, act_scales, layernorm_weights=None, layernorm_bias=None, alpha=0.5, weight_scales=None):
    """s[k] = act_absmax[k]^alpha / w_absmax[k]^(1 - alpha), clamped at 1e-5 (smoothquant.py:37-67; arXiv 2211.10438
    eq. 4).  `gemm_weights`: one or several [out, in] matrices sharing the input; they are multiplied by s along the
    input axis IN PLACE, `layernorm_weights` (the producer of the input) is divided by s in place.  Returns s."""
    if not isinstance(gemm_weights, (list, tuple)):
        gemm_weights = [gemm_weights]
    for g in gemm_weights:
        assert g.shape[1] == act_scales.numel(), 'weights are expected as [out, in]'
    if weight_scales is None:
        weight_scales = torch.stack([g.abs().amax(dim=0) for g in gemm_weights], dim=0).amax(dim=0)
    weight_scales = weight_scales.to(torch.float64).clamp(min=1e-5)
    scales = (act_scales.to(gemm_weights[0].device).to(torch.float64).pow(alpha) / weight_scales.pow(1 - alpha)).clamp(min=1e-5)
    if layernorm_weights is not None:
        layernorm_weights.div_(scales.to(layernorm_weights.dtype))
    if layernorm_bias is not None:
        layernorm_bias.div_(scales.to(layernorm_bias.dtype))
    for g in gemm_weights:
        g.mul_(scales.to(g.dtype).view(1, -1))
    return scales


@torch.no_grad()
def capture_activation_range(model, samples, num_samples=512):
    """Forward `samples` (an iterable of int64 token-id tensors [1, L]) through the HF model with hooks on every
    nn.Linear: per input channel max |x|, per output channel max |y|, per output channel max |w|
    (smoothquant.py:97-144)."""
    model.eval()
    device = next(model.parameters()).device
    act = defaultdict(lambda: {'x': None, 'y': None, 'w': None})

    def stat(name, t, key):
        m = t.reshape(-1, t.shape[-1]).abs().amax(dim=0).float()
        act[name][key] = m if act[name][key] is None else torch.maximum(act[name][key], m)

    def hook(mod, x, y, name):
        x = x[0] if isinstance(x, tuple) else x
        stat(name, x.detach(), 'x')
        stat(name, y.detach(), 'y')
        if act[name]['w'] is None:
            act[name]['w'] = mod.weight.detach().abs().clip(1e-8, None).amax(dim=1).float()

    hooks = [m.register_forward_hook(functools.partial(hook, name=n)) for n, m in model.named_modules()
             if isinstance(m, nn.Linear)]
    try:
        for i, ids in enumerate(samples):
            if i >= num_samples:
                break
            model(ids.to(device))
    finally:
        for h in hooks:
            h.remove()
    return act
