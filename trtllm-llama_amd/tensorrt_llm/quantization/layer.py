"""Quantised layers (T/tensorrt_llm/quantization/layer.py).

Weight-only:  WeightOnlyQuantLinear / WeightOnlyQuantRowLinear (layer.py:268-382) — weight Parameter is the
              fp32-typed [in, out/4] (int8) or [in, out/8] (int4) view of the processed bytes, scales fp16 [out].
SmoothQuant:  SmoothQuantLinear / SmoothQuantRowLinear (layer.py:70-220), and — designed for LLaMA by analogy to the
              GPT-2 classes (layer.py:223-265 norm+quant, :385-439 MLP, :596-852 attention):
              SmoothQuantRmsNorm, SmoothQuantGatedMLP, SmoothQuantAttention (RoPE kept inside the attention plugin,
              no bias, no query-key layer scaling)."""
from .._common import default_net
from ..functional import AttentionMaskType, PositionEmbeddingType, RaggedTensor, allgather, allreduce
from ..layers.attention import Attention
from ..module import Module
from ..parameter import Parameter
from .functional import (quantize_per_token, quantize_tensor, smooth_quant_gemm, smooth_quant_rms_norm,
                         weight_only_quant_matmul)
from .mode import QuantMode


class WeightOnlyQuantLinear(Module):

    def __init__(self, in_features, out_features, bias=False, dtype=None, tp_group=None, tp_size=1, gather_output=True,
                 quant_mode=QuantMode.use_weight_only()):
        super().__init__()
        if quant_mode.is_int8_weight_only():
            self.weight_only_quant_mode = 1
            quant_type_size_in_bits = 8
        elif quant_mode.is_int4_weight_only():
            self.weight_only_quant_mode = 2
            quant_type_size_in_bits = 4
        else:
            raise ValueError('WeightOnlyQuantLinear needs a weight-only QuantMode')
        self.in_features = in_features
        self.out_features = out_features // tp_size
        # 32 bits of the fp32 "port" carry 32 / bits quantised values (layer.py:289-292)
        self.weight = Parameter(shape=(self.in_features, int(self.out_features * quant_type_size_in_bits / 32)),
                                dtype='float32')
        self.per_channel_scale = Parameter(shape=(self.out_features, ), dtype=dtype)
        self.tp_size = tp_size
        self.tp_group = tp_group
        self.gather_output = gather_output
        if bias:
            raise NotImplementedError('bias is not built')
        self.register_parameter('bias', None)

    def forward(self, x):
        x = weight_only_quant_matmul(x, self.weight.value, self.per_channel_scale.value, self.weight_only_quant_mode)
        if self.gather_output and self.tp_size > 1 and self.tp_group is not None:
            x = allgather(x, self.tp_group)
        return x


WeightOnlyQuantColumnLinear = WeightOnlyQuantLinear


class WeightOnlyQuantRowLinear(Module):

    def __init__(self, in_features, out_features, bias=False, dtype=None, tp_group=None, tp_size=1,
                 quant_mode=QuantMode.use_weight_only()):
        super().__init__()
        if quant_mode.is_int8_weight_only():
            self.weight_only_quant_mode, bits = 1, 8
        elif quant_mode.is_int4_weight_only():
            self.weight_only_quant_mode, bits = 2, 4
        else:
            raise ValueError('WeightOnlyQuantRowLinear needs a weight-only QuantMode')
        self.in_features = in_features // tp_size
        self.out_features = out_features
        self.weight = Parameter(shape=(self.in_features, int(self.out_features * bits / 32)), dtype='float32')
        self.per_channel_scale = Parameter(shape=(self.out_features, ), dtype=dtype)
        if bias:
            raise NotImplementedError('bias is not built')
        self.register_parameter('bias', None)
        self.tp_group = tp_group
        self.tp_size = tp_size

    def forward(self, x):
        x = weight_only_quant_matmul(x, self.weight.value, self.per_channel_scale.value, self.weight_only_quant_mode)
        if self.tp_size > 1 and self.tp_group is not None:
            x = allreduce(x, self.tp_group)
        return x


class _SQLinearBase(Module):
    """int8 [out, in] weight smuggled as fp32 [out, in/4] (layer.py:91-99), per_channel_scale f32 [1, out] | [1, 1],
    act_scale f32 [1, 1] in static mode."""

    def _make(self, in_features, out_features, quant_mode):
        if not quant_mode.has_act_and_weight_quant():
            raise ValueError('SmoothQuant Linear has to have act+weight quantization mode set')
        self.in_features, self.out_features = in_features, out_features
        self.quant_mode = quant_mode
        self.weight = Parameter(shape=(out_features, in_features // 4), dtype='float32')
        n = out_features if quant_mode.has_per_channel_scaling() else 1
        self.per_channel_scale = Parameter(shape=(1, n), dtype='float32')
        if quant_mode.has_act_static_scaling():
            self.act_scale = Parameter(shape=(1, 1), dtype='float32')
        else:
            self.register_parameter('act_scale', None)
        self.register_parameter('bias', None)

    def _gemm(self, x):
        if self.quant_mode.has_act_static_scaling():
            per_token_scale = self.act_scale.value
        else:
            x, per_token_scale = x  # (int8 activations, per-token scales)
        # the plugin reads scales_b as a vector only when told so: a loader that expands "per tensor" to one factor each for
        # Q, K and V ([1, 3 * out / tp], examples/llama_quant/weight.py) must get per-channel dequantisation for that GEMM
        # even though the model's QuantMode says per-tensor - decided from the tensor the engine will actually carry
        per_channel = self.quant_mode.has_per_channel_scaling() or int(tuple(self.per_channel_scale.shape)[-1]) > 1
        return smooth_quant_gemm(x, self.weight.value, per_token_scale, self.per_channel_scale.value,
                                 self.quant_mode.has_per_token_dynamic_scaling(), per_channel)


class SmoothQuantLinear(_SQLinearBase):

    def __init__(self, in_features, out_features, bias=False, dtype=None, tp_group=None, tp_size=1, gather_output=True,
                 quant_mode=QuantMode(0)):
        super().__init__()
        if bias:
            raise NotImplementedError('bias is not built')
        self._make(in_features, out_features // tp_size, quant_mode)
        self.dtype = dtype
        self.tp_size, self.tp_group, self.gather_output = tp_size, tp_group, gather_output

    def forward(self, x):
        x = self._gemm(x)
        if self.gather_output and self.tp_size > 1 and self.tp_group is not None:
            x = allgather(x, self.tp_group)
        return x


SmoothQuantColumnLinear = SmoothQuantLinear


class SmoothQuantRowLinear(_SQLinearBase):

    def __init__(self, in_features, out_features, bias=False, dtype=None, tp_group=None, tp_size=1,
                 quant_mode=QuantMode(0)):
        super().__init__()
        if bias:
            raise NotImplementedError('bias is not built')
        self._make(in_features // tp_size, out_features, quant_mode)
        self.dtype = dtype
        self.tp_size, self.tp_group = tp_size, tp_group

    def forward(self, x):
        x = self._gemm(x)
        if self.tp_size > 1 and self.tp_group is not None:
            x = allreduce(x, self.tp_group)
        return x


class SmoothQuantRmsNorm(Module):
    """RMSNorm -> int8 (static `scale_to_int`, or per-token dynamic: returns (int8, scales))."""

    def __init__(self, normalized_shape, eps=1e-06, elementwise_affine=True, dtype=None, quant_mode=QuantMode(0)):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape, )
        if not quant_mode.has_act_and_weight_quant():
            raise ValueError('SmoothQuant RMS norm has to have some quantization mode set')
        self.normalized_shape = tuple(normalized_shape)
        self.quant_mode = quant_mode
        self.weight = Parameter(shape=self.normalized_shape, dtype=dtype)
        self.eps = eps
        if quant_mode.has_act_static_scaling():
            self.scale_to_int = Parameter(shape=(1, ), dtype='float32')
        else:
            self.register_parameter('scale_to_int', None)

    def forward(self, x):
        scale = self.scale_to_int.value if self.scale_to_int is not None else self.weight.value  # dummy when dynamic
        return smooth_quant_rms_norm(x, self.normalized_shape, self.weight.value, scale, self.eps,
                                     dynamic_act_scaling=self.quant_mode.has_per_token_dynamic_scaling())


class SmoothQuantGatedMLP(Module):
    """int8 fc | gate (shared int8 input) -> fp16 -> silu(fc) * gate -> quantise -> int8 proj -> fp16."""

    def __init__(self, hidden_size, ffn_hidden_size, hidden_act, bias=False, dtype=None, tp_group=None, tp_size=1,
                 quant_mode=QuantMode(0)):
        super().__init__()
        from ..functional import ACT2FN
        if hidden_act not in ACT2FN:
            raise ValueError(f'unsupported activation function: {hidden_act}')
        self.fc = SmoothQuantLinear(hidden_size, ffn_hidden_size, bias=bias, dtype=dtype, tp_group=tp_group,
                                    tp_size=tp_size, gather_output=False, quant_mode=quant_mode)
        self.gate = SmoothQuantLinear(hidden_size, ffn_hidden_size, bias=bias, dtype=dtype, tp_group=tp_group,
                                      tp_size=tp_size, gather_output=False, quant_mode=quant_mode)
        self.proj = SmoothQuantRowLinear(ffn_hidden_size, hidden_size, bias=bias, dtype=dtype, tp_group=tp_group,
                                         tp_size=tp_size, quant_mode=quant_mode)
        self.hidden_act = hidden_act
        self.quant_mode = quant_mode
        self.dtype = dtype
        if quant_mode.has_act_static_scaling():
            self.quantization_scaling_factor = Parameter(shape=(1, ), dtype='float32')
        else:
            self.register_parameter('quantization_scaling_factor', None)

    def forward(self, hidden_states):
        from ..functional import ACT2FN
        inter = ACT2FN[self.hidden_act](self.fc(hidden_states)) * self.gate(hidden_states)
        if self.quant_mode.has_act_static_scaling():
            inter = quantize_tensor(inter, self.quantization_scaling_factor.value)
        else:
            inter = quantize_per_token(inter)
        return self.proj(inter)


class SmoothQuantAttention(Attention):
    """int8 QKV GEMM -> fp16 -> GPTAttention plugin (RoPE, fp16|int8 KV) -> quantise context -> int8 dense GEMM."""

    def __init__(self, hidden_size, num_attention_heads, max_position_embeddings, num_layers=1,
                 apply_query_key_layer_scaling=False, attention_mask_type=AttentionMaskType.causal, bias=False,
                 dtype=None, position_embedding_type=PositionEmbeddingType.rope, neox_rotary_style=True,
                 use_int8_kv_cache=False, tp_group=None, tp_size=1, multi_block_mode=False, multi_query_mode=False,
                 quant_mode=QuantMode(0)):
        super().__init__(hidden_size, num_attention_heads, max_position_embeddings, num_layers,
                         apply_query_key_layer_scaling, attention_mask_type, False, dtype, position_embedding_type,
                         neox_rotary_style, use_int8_kv_cache, 1.0, tp_group, tp_size, multi_block_mode,
                         multi_query_mode)
        self.quant_mode = quant_mode
        self.qkv = SmoothQuantLinear(hidden_size, hidden_size * 3, bias=bias, dtype=dtype, tp_group=tp_group,
                                     tp_size=tp_size, gather_output=False, quant_mode=quant_mode)
        self.dense = SmoothQuantRowLinear(hidden_size, hidden_size, bias=bias, dtype=dtype, tp_group=tp_group,
                                          tp_size=tp_size, quant_mode=quant_mode)
        if quant_mode.has_act_static_scaling():
            self.quantization_scaling_factor = Parameter(shape=(1, ), dtype='float32')
        else:
            self.register_parameter('quantization_scaling_factor', None)

    def forward(self, hidden_states: RaggedTensor, attention_mask=None, past_key_value=None, sequence_length=None,
                past_key_value_length=None, masked_tokens=None, use_cache=False, cache_indirection=None, **kwargs):
        assert isinstance(hidden_states, RaggedTensor)
        if not default_net().plugin_config.smooth_quant_gemm_plugin:
            raise ValueError('smooth_quant_gemm_plugin is not set')
        qkv = self.qkv(hidden_states.data)  # .data is int8 (static) or (int8, scales) (dynamic)
        context, present = self._attend(qkv, hidden_states, past_key_value, sequence_length, past_key_value_length,
                                        masked_tokens, cache_indirection)
        if self.quant_mode.has_act_static_scaling():
            context = quantize_tensor(context, self.quantization_scaling_factor.value)
        else:
            context = quantize_per_token(context)
        context = self.dense(context)
        context = RaggedTensor.from_row_lengths(context, hidden_states.row_lengths, hidden_states.max_row_length)
        return (context, present) if use_cache else context
