"""Module surgery that turns the fp16 LLaMA into its quantised forms (T/tensorrt_llm/models/quantized/quant.py,
T/examples/llama_quant/quant.py:8-75).

smooth_quantize: the reference wires LayerNorm + a RoPE-less, biased attention and leaves the GatedMLP unquantised
(quant.py:15-40; its engine build fails, README.md:856).  Here the LLaMA block of SURVEY Appendix A.4 is built:
SmoothQuantRmsNorm, SmoothQuantAttention (RoPE in the plugin, no bias, q_scaling 1), SmoothQuantGatedMLP.
weight_only_quantize: every ColumnLinear/RowLinear except lm_head (quant.py:52-75)."""
from ...layers import ColumnLinear, RowLinear
from ...quantization.layer import (SmoothQuantAttention, SmoothQuantGatedMLP, SmoothQuantRmsNorm,
                                   WeightOnlyQuantColumnLinear, WeightOnlyQuantRowLinear)


def smooth_quantize(model, quant_mode):
    assert quant_mode.has_act_and_weight_quant()
    for layer in model.layers:
        assert hasattr(layer, 'input_layernorm'), 'The layer has no input_layernorm'
        layer.input_layernorm = SmoothQuantRmsNorm(normalized_shape=layer.hidden_size, dtype=layer.dtype,
                                                   quant_mode=quant_mode)
        assert hasattr(layer, 'attention'), 'The layer has no attention'
        layer.attention = SmoothQuantAttention(layer.hidden_size, layer.num_attention_heads,
                                               layer.max_position_embeddings, dtype=layer.dtype,
                                               attention_mask_type=layer.attention_mask_type,
                                               position_embedding_type=layer.position_embedding_type,
                                               neox_rotary_style=layer.neox_rotary_style,
                                               use_int8_kv_cache=quant_mode.has_int8_kv_cache(),
                                               tp_group=layer.tp_group, tp_size=layer.tp_size, quant_mode=quant_mode)
        assert hasattr(layer, 'mlp'), 'The layer has no mlp'
        layer.mlp = SmoothQuantGatedMLP(hidden_size=layer.hidden_size, ffn_hidden_size=layer.mlp_hidden_size,
                                        hidden_act=layer.hidden_act, dtype=layer.dtype, tp_group=layer.tp_group,
                                        tp_size=layer.tp_size, quant_mode=quant_mode)
        assert hasattr(layer, 'post_layernorm'), 'The layer has no post_layernorm'
        layer.post_layernorm = SmoothQuantRmsNorm(normalized_shape=layer.hidden_size, dtype=layer.dtype,
                                                  quant_mode=quant_mode)
    setattr(model, 'quant_mode', quant_mode)
    return model


def weight_only_quantize(model, quant_mode, exclude_modules=None, current_key_name=None):
    assert quant_mode.is_weight_only()
    exclude_modules = ['lm_head'] if exclude_modules is None else exclude_modules
    for name, module in model.named_children():
        path = (current_key_name or []) + [name]
        if module is None:
            continue
        if len(module.children()) > 0:
            weight_only_quantize(module, quant_mode, exclude_modules, path)
        if name in exclude_modules or any(key in '.'.join(path) for key in exclude_modules):
            continue
        if isinstance(module, ColumnLinear):
            model._modules[name] = WeightOnlyQuantColumnLinear(
                in_features=module.in_features, out_features=module.out_features * module.tp_size,
                bias=module.bias is not None, dtype=module.dtype, tp_group=module.tp_group, tp_size=module.tp_size,
                gather_output=module.gather_output, quant_mode=quant_mode)
        elif isinstance(module, RowLinear):
            model._modules[name] = WeightOnlyQuantRowLinear(
                in_features=module.in_features * module.tp_size, out_features=module.out_features,
                bias=module.bias is not None, dtype=module.dtype, tp_group=module.tp_group, tp_size=module.tp_size,
                quant_mode=quant_mode)
    setattr(model, 'quant_mode', quant_mode)
    return model
