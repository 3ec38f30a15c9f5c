"""functional: the tensor ops the LLaMA layer is written in (subset of T/tensorrt_llm/functional.py — SURVEY §2.1 row 7).

"Define-and-run" like the reference: every call appends one node to the default Network.  Where the reference adds
a TensorRT plugin layer (`trt.get_plugin_registry().get_plugin_creator(name, '1', 'tensorrt_llm')` +
`create_plugin(PluginFieldCollection)`), the node is created through the same-shaped C ABI
(`tllm_plugin_create(name, '1', 'tensorrt_llm', fields)`), so a wrong / missing field fails at trace time exactly
where the reference's creator would return None.
"""
from collections import OrderedDict
from enum import IntEnum
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from ._common import default_net, has_default_net
from ._utils import DataType, str_dtype_to_trt, trt_dtype_to_str
from .plugin import TRT_LLM_PLUGIN_NAMESPACE, capi


class DimRange(object):
    """min/opt/max ranges of dynamic dims (kept for prepare_inputs() source compatibility)."""

    def __init__(self, shape: List[Union[int, List[int], Tuple[int, int, int]]]):
        self.min, self.opt, self.max = [], [], []
        for dim in shape:
            if isinstance(dim, (list, tuple)):
                assert len(dim) == 3
                self.min.append(dim[0])
                self.opt.append(dim[1])
                self.max.append(dim[2])
            else:
                self.min.append(dim)
                self.opt.append(dim)
                self.max.append(dim)


class Tensor(object):
    """Symbolic tensor of the traced network (dims may be -1 = dynamic)."""

    def __init__(self, name=None, dtype=None, shape=None, dim_range=None, is_network_input=True, network=None):
        if isinstance(dtype, str):
            dtype = str_dtype_to_trt(dtype)
        self.network = network if network is not None else (default_net() if has_default_net() else None)
        self.name = name if name is not None else (self.network.new_name() if self.network else 'tensor')
        self.dtype = DataType(dtype) if dtype is not None else None
        self.shape = tuple(shape) if shape is not None else None
        self.dim_range = dim_range
        self.producer = None
        if is_network_input and self.network is not None and name is not None:
            self.network.add_input(self)

    # the reference exposes the wrapped trt.ITensor; layers only pass it back into functional ops
    @property
    def trt_tensor(self):
        return self

    def mark_output(self, name, dtype=None):
        if isinstance(dtype, str):
            dtype = str_dtype_to_trt(dtype)
        self.network.mark_output(self, name, dtype if dtype is not None else self.dtype)

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def ndim(self):
        return len(self.shape)

    def rank(self):
        return len(self.shape)

    def __add__(self, other):
        return add(self, other)

    def __radd__(self, other):
        return add(other, self)

    def __mul__(self, other):
        return mul(self, other)

    def __rmul__(self, other):
        return mul(other, self)

    def __repr__(self):
        return f'Tensor({self.name}, {trt_dtype_to_str(self.dtype) if self.dtype is not None else None}, {self.shape})'


def _new(dtype, shape, hint='t') -> Tensor:
    net = default_net()
    return Tensor(name=net.new_name(hint), dtype=dtype, shape=shape, is_network_input=False, network=net)


def _create_tensor(t: Tensor, producer=None) -> Tensor:
    t.producer = producer
    return t


class RaggedTensor(object):
    """A padded [batch, max_len, ...] tensor plus per-row lengths (T/tensorrt_llm/functional.py:351-413)."""

    def __init__(self):
        self._data = None
        self._row_lengths = None
        self._max_row_length = None

    @staticmethod
    def from_row_lengths(data: Tensor, row_lengths: Tensor, max_row_length: Tensor = None) -> 'RaggedTensor':
        r = RaggedTensor()
        r._data, r._row_lengths, r._max_row_length = data, row_lengths, max_row_length
        return r

    @property
    def data(self) -> Tensor:
        return self._data

    @property
    def row_lengths(self) -> Tensor:
        return self._row_lengths

    @property
    def max_row_length(self) -> Tensor:
        return self._max_row_length


class AttentionMaskType(IntEnum):
    padding = 0
    causal = 1
    bidirectional = 2


class PositionEmbeddingType(IntEnum):
    learned_absolute = 0
    rope = 1
    alibi = 2


# ------------------------------------------------------------------------------------------------ basic ops
def constant(value: np.ndarray, parameter=None, name_hint='const') -> Tensor:
    """A weight entering the graph.  `parameter` ties the constant to a Module Parameter so that the builder can
    store it under its module path."""
    net = default_net()
    v = np.asarray(value) if parameter is None else None
    shape = tuple(parameter.shape) if parameter is not None else tuple(v.shape)
    dtype = parameter.dtype if parameter is not None else {np.dtype(np.float32): DataType.FLOAT,
                                                           np.dtype(np.float16): DataType.HALF,
                                                           np.dtype(np.int32): DataType.INT32,
                                                           np.dtype(np.int8): DataType.INT8}[v.dtype]
    t = _new(dtype, shape, name_hint)
    net._constants.append((t.name, parameter if parameter is not None else v))
    net.add_node('constant', [], [t], shape=list(shape), dtype=int(dtype))
    return t


def _bshape(a: Sequence[int], b: Sequence[int]):
    out = []
    for x, y in zip(([1] * (len(b) - len(a)) + list(a)), ([1] * (len(a) - len(b)) + list(b))):
        out.append(-1 if (x == -1 or y == -1) else max(x, y))
    return tuple(out)


def _as_tensor(x, like: Tensor) -> Tensor:
    if isinstance(x, Tensor):
        return x
    return constant(np.array(x, dtype={DataType.HALF: np.float16, DataType.FLOAT: np.float32,
                                       DataType.INT32: np.int32}[like.dtype]))


def add(a, b) -> Tensor:
    ref = a if isinstance(a, Tensor) else b
    a, b = _as_tensor(a, ref), _as_tensor(b, ref)
    out = _new(a.dtype, _bshape(a.shape, b.shape), 'add')
    default_net().add_node('add', [a, b], [out])
    return out


def mul(a, b) -> Tensor:
    ref = a if isinstance(a, Tensor) else b
    a, b = _as_tensor(a, ref), _as_tensor(b, ref)
    out = _new(a.dtype, _bshape(a.shape, b.shape), 'mul')
    default_net().add_node('mul', [a, b], [out])
    return out


def silu(x: Tensor) -> Tensor:
    """x * sigmoid(x) (T/tensorrt_llm/functional.py:521-532)."""
    out = _new(x.dtype, x.shape, 'silu')
    default_net().add_node('silu', [x], [out])
    return out


def swiglu(x: Tensor) -> Tensor:
    """silu(x[..., :n]) * x[..., n:] (functional.py:535-551)."""
    n = x.shape[-1] // 2 if x.shape[-1] != -1 else -1
    out = _new(x.dtype, tuple(x.shape[:-1]) + (n, ), 'swiglu')
    default_net().add_node('swiglu', [x], [out])
    return out


ACT2FN = {'silu': silu, 'swiglu': swiglu}


def cast(x: Tensor, dtype) -> Tensor:
    if isinstance(dtype, str):
        dtype = str_dtype_to_trt(dtype)
    out = _new(dtype, x.shape, 'cast')
    default_net().add_node('cast', [x], [out], dtype=int(dtype))
    return out


def shape(x: Tensor, dim: Optional[int] = None) -> Tensor:
    out = _new(DataType.INT32, (len(x.shape), ) if dim is None else (), 'shape')
    default_net().add_node('shape', [x], [out], dim=dim)
    return out


def assertion(condition, message: str = '') -> None:
    default_net().add_node('assertion', [condition] if isinstance(condition, Tensor) else [], [], message=message)


def expand_mask(mask: Tensor, tgt_len=None) -> Tensor:
    """[batch, src] -> [batch, 1, tgt, src] additive mask; the attention plugin ignores it (the reference feeds it
    only to the non-plugin attention path)."""
    out = _new(mask.dtype, (mask.shape[0], 1, -1, mask.shape[-1]), 'mask')
    default_net().add_node('expand_mask', [mask], [out])
    return out


def embedding(input: Tensor, weight: Tensor) -> Tensor:
    """Token gather from the [vocab, hidden] table (functional.py:1642-1703; plain gather, no lookup plugin)."""
    out = _new(weight.dtype, tuple(input.shape) + (weight.shape[-1], ), 'embedding')
    default_net().add_node('embedding', [input, weight], [out])
    return out


def matmul(input: Tensor, mat2: Tensor, transa: bool = False, transb: bool = False) -> Tensor:
    n = mat2.shape[-2] if transb else mat2.shape[-1]
    out = _new(input.dtype, tuple(input.shape[:-1]) + (n, ), 'matmul')
    default_net().add_node('matmul', [input, mat2], [out], transa=bool(transa), transb=bool(transb))
    return out


def rms_norm(input: Tensor, normalized_shape, weight: Optional[Tensor] = None, eps: float = 1e-06) -> Tensor:
    """y = x / sqrt(mean(x^2) + eps) * w, statistics in fp32 (functional.py:3195-3219).  One node here: the reference
    composes it from pow/mean/add/sqrt/div/mul TensorRT layers which Myelin fuses; the engine runs a real kernel."""
    out = _new(input.dtype, input.shape, 'rmsnorm')
    ins = [input] + ([weight] if weight is not None else [])
    default_net().add_node('rms_norm', ins, [out], eps=float(eps),
                           normalized_shape=list(normalized_shape) if isinstance(normalized_shape, (list, tuple)) else [
                               int(normalized_shape)])
    return out


def gather_last_token_logits(hidden_states: Tensor, last_token_ids: Tensor, remove_input_padding: bool) -> Tensor:
    """hidden[b, last_token_ids[b] - 1, :] (functional.py:3316-3380)."""
    # packed inputs: hidden is [1, num_tokens, H] and last_token_ids holds inclusive prefix sums of the lengths
    batch = last_token_ids.shape[0] if remove_input_padding else hidden_states.shape[0]
    out = _new(hidden_states.dtype, (batch, hidden_states.shape[-1]), 'last_token')
    default_net().add_node('gather_last_token_logits', [hidden_states, last_token_ids], [out],
                           remove_input_padding=bool(remove_input_padding))
    return out


# ------------------------------------------------------------------------------------------------ plugin nodes
def _field(name, value, np_dtype):
    return capi.PluginField(name, np.array(value, dtype=np_dtype))


def _add_plugin(plugin_name: str, fields: List[capi.PluginField], inputs: List[Tensor], layer_name: str):
    """creator lookup + create_plugin + add_plugin_v2 of the reference, through the C ABI."""
    plug = capi.Plugin.create(plugin_name, fields, '1', TRT_LLM_PLUGIN_NAMESPACE)
    assert plug is not None, f'{plugin_name} plugin creation failed: {capi.last_error()}'
    static = all(all(d >= 0 for d in t.shape) for t in inputs)
    outs = []
    for i in range(plug.num_outputs):
        dt = DataType(plug.output_dtype(i, [int(t.dtype) for t in inputs]))
        shp = tuple(plug.output_dims(i, [list(t.shape) for t in inputs])) if static else None
        outs.append(_new(dt, shp, plugin_name.lower()))
    default_net().add_node('plugin', inputs, outs, plugin_type=plugin_name, layer_name=layer_name,
                           fields=OrderedDict((f.name, f.data.tolist()) for f in fields),
                           serialized=plug.serialize().hex())
    plug.destroy()
    return outs


def _dyn_like(t: Tensor, last: int):
    return tuple(t.shape[:-1]) + (last, )


def gemm_plugin(input: Tensor, mat2: Tensor, transa: bool = False, transb: bool = False) -> Tensor:
    """`_gemm_plugin` of T/tensorrt_llm/layers/linear.py:13-35: fields transa, transb, type_id."""
    p_dtype = default_net().plugin_config.gemm_plugin
    fields = [_field('transa', 1 if transa else 0, np.int32), _field('transb', 1 if transb else 0, np.int32),
              _field('type_id', [int(str_dtype_to_trt(p_dtype))], np.int32)]
    out, = _add_plugin('Gemm', fields, [input, mat2], 'gemm')
    if out.shape is None:
        out.shape = _dyn_like(input, mat2.shape[0] if transb else mat2.shape[1])
    return out


def gpt_attention(tensor: Tensor, past_key_value: Tensor, sequence_length: Tensor, past_key_value_length: Tensor,
                  masked_tokens: Tensor, input_lengths: Tensor, max_input_length: Tensor, cache_indirection: Tensor,
                  num_heads: int, head_size: int, q_scaling: float, rotary_embedding_dim: int,
                  neox_rotary_style: bool, multi_block_mode: bool, multi_query_mode: bool,
                  kv_orig_quant_scale: Tensor = None, kv_quant_orig_scale: Tensor = None,
                  use_int8_kv_cache: bool = False, use_fp8_kv_cache: bool = False,
                  mask_type: int = int(AttentionMaskType.causal), kv_cache_block_pointers: Tensor = None,
                  host_input_lengths: Tensor = None, host_request_types: Tensor = None) -> Tuple[Tensor, Tensor]:
    """The GPTAttention plugin node (T/tensorrt_llm/functional.py:2695-2928): same 16 fields, same input order
    (tensor, past_key_value, sequence_length, past_key_value_length [host], masked_tokens, input_lengths,
    max_input_length, cache_indirection [, kv_orig_quant_scale, kv_quant_orig_scale])."""
    cfg = default_net().plugin_config
    assert head_size in [32, 48, 64, 80, 96, 128, 144, 160, 192, 224, 256]
    p_dtype = cfg.gpt_attention_plugin
    assert p_dtype, 'gpt_attention requires plugin_config.set_gpt_attention_plugin()'
    fields = [
        _field('num_heads', num_heads, np.int32), _field('head_size', head_size, np.int32),
        _field('unidirectional', 1, np.int32), _field('q_scaling', q_scaling, np.float32),
        _field('rotary_embedding_dim', rotary_embedding_dim, np.int32),
        _field('neox_rotary_style', 1 if neox_rotary_style else 0, np.int8),
        _field('context_fmha_type', int(cfg.context_fmha_type), np.int8),
        _field('multi_block_mode', 1 if multi_block_mode else 0, np.int8),
        _field('multi_query_mode', 1 if multi_query_mode else 0, np.int8),
        _field('int8_kv_cache', 1 if use_int8_kv_cache else 0, np.int32),
        _field('fp8_kv_cache', 1 if use_fp8_kv_cache else 0, np.int32),
        _field('remove_input_padding', 1 if cfg.remove_input_padding else 0, np.int8),
        _field('mask_type', [int(mask_type)], np.int32), _field('paged_kv_cache', 1 if cfg.paged_kv_cache else 0, np.int32),
        _field('type_id', [int(str_dtype_to_trt(p_dtype))], np.int32),
        _field('in_flight_batching', 1 if cfg.in_flight_batching else 0, np.int32),
    ]
    plug_inputs = [tensor, past_key_value, sequence_length, past_key_value_length, masked_tokens, input_lengths,
                   max_input_length, cache_indirection]
    if use_int8_kv_cache or use_fp8_kv_cache:
        plug_inputs += [kv_orig_quant_scale, kv_quant_orig_scale]
    if cfg.paged_kv_cache:
        plug_inputs += [kv_cache_block_pointers]
    if cfg.in_flight_batching:
        plug_inputs += [host_input_lengths, host_request_types]
    output, present = _add_plugin('GPTAttention', fields, plug_inputs, 'causal_attn')
    if output.shape is None:
        output.shape = _dyn_like(tensor, num_heads * head_size)
        present.shape = past_key_value.shape
    return output, present


def _collective(name: str, tensor: Tensor, group: List[int]) -> Tensor:
    p_dtype = default_net().plugin_config.nccl_plugin or trt_dtype_to_str(tensor.dtype)
    fields = [_field('group', list(group), np.int32), _field('type_id', [int(str_dtype_to_trt(p_dtype))], np.int32)]
    out, = _add_plugin(name, fields, [tensor], name.lower())
    if out.shape is None:
        out.shape = tensor.shape if name == 'AllReduce' else ((-1, ) + tuple(tensor.shape[1:]))
    return out


def allreduce(tensor: Tensor, group: List[int]) -> Tensor:
    """sum-all-reduce over the TP group (functional.py:2422-2470 -> P/ncclPlugin/allreducePlugin.cpp)."""
    return _collective('AllReduce', tensor, group)


def allgather(tensor: Tensor, group: List[int]) -> Tensor:
    """all-gather along dim 0 (functional.py:2473-2522 -> P/ncclPlugin/allgatherPlugin.cpp)."""
    return _collective('AllGather', tensor, group)
