"""Quantisation plugin wrappers (T/tensorrt_llm/quantization/functional.py): smooth_quant_gemm,
weight_only_quant_matmul, quantize_tensor, quantize_per_token, and smooth_quant_rms_norm — the RMSNorm analogue of
smooth_quant_layer_norm that LLaMA needs (SURVEY "fact 1").  Same plugin names / field names / input order."""
from typing import Tuple

import numpy as np

from .._common import default_net
from .._utils import str_dtype_to_trt
from ..functional import Tensor, _add_plugin, _dyn_like, _field


def smooth_quant_gemm(input: Tensor, weights: Tensor, scales_a: Tensor, scales_b: Tensor, per_token_scaling: bool,
                      per_channel_scaling: bool) -> Tensor:
    """SmoothQuantGemm: int8 x int8 -> int32, * (scale_a[m] * scale_b[n]) (quantization/functional.py:12-48)."""
    p_dtype = default_net().plugin_config.smooth_quant_gemm_plugin
    if not p_dtype:
        raise TypeError('Smooth Quant GEMM is only supported with plugin')
    fields = [_field('has_per_channel_scaling', 1 if per_channel_scaling else 0, np.int32),
              _field('has_per_token_scaling', 1 if per_token_scaling else 0, np.int32),
              _field('type_id', [int(str_dtype_to_trt(p_dtype))], np.int32)]
    out, = _add_plugin('SmoothQuantGemm', fields, [input, weights, scales_a, scales_b], 'sq_gemm')
    if out.shape is None:
        out.shape = _dyn_like(input, weights.shape[0])
    return out


def weight_only_quant_matmul(input: Tensor, weights: Tensor, scales: Tensor, weightTypeId: int) -> Tensor:
    """WeightOnlyQuantMatmul: fp16 x (int8|int4 * scale) (quantization/functional.py:51-74)."""
    p_dtype = default_net().plugin_config.weight_only_quant_matmul_plugin
    if not p_dtype:
        raise TypeError('Weight Only Qunat MatMul is only supported with plugin')
    fields = [_field('type_id', [int(str_dtype_to_trt(p_dtype))], np.int32),
              _field('weight_type_id', weightTypeId, np.int32)]
    out, = _add_plugin('WeightOnlyQuantMatmul', fields, [input, weights, scales], 'woq_matmul')
    if out.shape is None:
        out.shape = _dyn_like(input, scales.shape[-1])
    return out


def smooth_quant_rms_norm(input: Tensor, normalized_shape, weight: Tensor, scale: Tensor = None, eps: float = 1e-06,
                          dynamic_act_scaling: bool = False):
    """RmsnormQuantization: RMSNorm fused with int8 quantisation (static per-tensor `scale`, or per-token dynamic
    scales returned as a second output) — pattern of smooth_quant_layer_norm (quantization/functional.py:77-129)."""
    cfg = default_net().plugin_config
    p_dtype = cfg.rmsnorm_quantization_plugin or cfg.layernorm_quantization_plugin
    if not p_dtype:
        raise TypeError('Smooth Quant RMS Norm is only supported with plugin')
    fields = [_field('eps', eps, np.float32), _field('dyn_act_scaling', [1 if dynamic_act_scaling else 0], np.int32),
              _field('type_id', [int(str_dtype_to_trt(p_dtype))], np.int32)]
    outs = _add_plugin('RmsnormQuantization', fields, [input, weight, scale], 'rmsnorm_quantized')
    if outs[0].shape is None:
        outs[0].shape = input.shape
        if dynamic_act_scaling:
            outs[1].shape = _dyn_like(input, 1)
    return tuple(outs) if dynamic_act_scaling else outs[0]


def quantize_per_token(x: Tensor) -> Tuple[Tensor, Tensor]:
    """QuantizePerToken: (int8 [.., K], f32 scales [.., 1]) (quantization/functional.py:160-189)."""
    if not default_net().plugin_config.quantize_per_token_plugin:
        raise TypeError('quantize_per_token is only built as a plugin on MI355X')
    q, s = _add_plugin('QuantizePerToken', [], [x], 'quantize_per_token_plugin')
    if q.shape is None:
        q.shape, s.shape = x.shape, _dyn_like(x, 1)
    return q, s


def quantize_tensor(x: Tensor, scale: Tensor) -> Tensor:
    """QuantizeTensor: int8 = sat(rni(x * scale)) (quantization/functional.py:192-212)."""
    if not default_net().plugin_config.quantize_tensor_plugin:
        raise TypeError('quantize_tensor is only built as a plugin on MI355X')
    q, = _add_plugin('QuantizeTensor', [], [x, scale], 'quantize_tensor_plugin')
    if q.shape is None:
        q.shape = x.shape
    return q
