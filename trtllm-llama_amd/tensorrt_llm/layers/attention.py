"""Attention (T/tensorrt_llm/layers/attention.py:48-184): fused QKV ColumnLinear -> GPTAttention plugin (RoPE, KV cache,
masked MHA) -> dense RowLinear.  Only the plugin path exists on MI355X (RoPE requires it in the reference too,
attention.py:141-144)."""
import math

from .._common import default_net
from ..functional import AttentionMaskType, PositionEmbeddingType, RaggedTensor, gpt_attention
from ..module import Module
from ..parameter import Parameter
from .linear import ColumnLinear, RowLinear


class Attention(Module):

    def __init__(self, hidden_size, num_attention_heads, max_position_embeddings, num_layers=1,
                 apply_query_key_layer_scaling=False, attention_mask_type=AttentionMaskType.padding, bias=True,
                 dtype=None, position_embedding_type=PositionEmbeddingType.learned_absolute, neox_rotary_style=False,
                 use_int8_kv_cache=False, rotary_embedding_percentage=1.0, tp_group=None, tp_size=1,
                 multi_block_mode=False, multi_query_mode=False):
        super().__init__()
        self.attention_mask_type = attention_mask_type
        self.attention_head_size = hidden_size // num_attention_heads
        self.num_attention_heads = num_attention_heads // tp_size
        self.num_attention_kv_heads = 1 if multi_query_mode else self.num_attention_heads
        self.hidden_size = hidden_size // tp_size
        self.max_position_embeddings = max_position_embeddings
        self.num_layers = num_layers
        self.apply_query_key_layer_scaling = apply_query_key_layer_scaling
        self.norm_factor = math.sqrt(self.attention_head_size)
        self.q_scaling = 1
        if apply_query_key_layer_scaling:  # attention.py:79-82
            self.norm_factor *= num_layers
            self.q_scaling *= num_layers
        self.position_embedding_type = position_embedding_type
        self.multi_block_mode = multi_block_mode
        self.multi_query_mode = multi_query_mode
        self.neox_rotary_style = neox_rotary_style
        self.rotary_embedding_dim = 0
        if position_embedding_type == PositionEmbeddingType.rope:
            self.rotary_embedding_dim = int(self.attention_head_size * rotary_embedding_percentage)  # :88-92
        self.dtype = dtype
        self.use_int8_kv_cache = use_int8_kv_cache
        if use_int8_kv_cache:  # :98-104
            self.kv_orig_quant_scale = Parameter(shape=(1, ), dtype='float32')
            self.kv_quant_orig_scale = Parameter(shape=(1, ), dtype='float32')
        else:
            self.register_parameter('kv_orig_quant_scale', None)
            self.register_parameter('kv_quant_orig_scale', None)
        if multi_query_mode:
            raise NotImplementedError('multi_query_mode is not built (LLaMA-7B is MHA)')
        self.qkv = ColumnLinear(hidden_size, hidden_size * 3, bias=bias, dtype=dtype, tp_group=tp_group, tp_size=tp_size,
                                gather_output=False)
        self.dense = RowLinear(hidden_size, hidden_size, bias=bias, dtype=dtype, tp_group=tp_group, tp_size=tp_size)

    def _attend(self, qkv, hidden_states: RaggedTensor, past_key_value, sequence_length, past_key_value_length,
                masked_tokens, cache_indirection, kv_cache_block_pointers=None):
        cfg = default_net().plugin_config
        if not cfg.gpt_attention_plugin:
            raise ValueError('RoPE is only supported with GPTAttention plugin'
                             if self.position_embedding_type == PositionEmbeddingType.rope else
                             'only the GPTAttention plugin path is built for MI355X')
        assert sequence_length is not None and past_key_value_length is not None
        assert masked_tokens is not None and cache_indirection is not None
        assert self.attention_mask_type in (AttentionMaskType.causal, AttentionMaskType.bidirectional), \
            'Plugin only support masked MHA.'
        assert hidden_states.row_lengths is not None
        if self.position_embedding_type == PositionEmbeddingType.alibi:
            raise ValueError('ALiBi is only supported without GPTAttention plugin')
        kv_oq = self.kv_orig_quant_scale.value if self.use_int8_kv_cache else None
        kv_qo = self.kv_quant_orig_scale.value if self.use_int8_kv_cache else None
        return gpt_attention(qkv, past_key_value, sequence_length, past_key_value_length, masked_tokens,
                             hidden_states.row_lengths, hidden_states.max_row_length, cache_indirection,
                             self.num_attention_heads, self.attention_head_size, self.q_scaling,
                             self.rotary_embedding_dim, self.neox_rotary_style, self.multi_block_mode,
                             self.multi_query_mode, kv_oq, kv_qo, self.use_int8_kv_cache,
                             kv_cache_block_pointers=kv_cache_block_pointers)

    def forward(self, hidden_states: RaggedTensor, attention_mask=None, past_key_value=None, sequence_length=None,
                past_key_value_length=None, masked_tokens=None, use_cache=False, cache_indirection=None,
                kv_cache_block_pointers=None, inflight_batching_args=None, past_key_value_pointers=None):
        assert isinstance(hidden_states, RaggedTensor)
        qkv = self.qkv(hidden_states.data)
        context, present = self._attend(qkv, hidden_states, past_key_value, sequence_length, past_key_value_length,
                                        masked_tokens, cache_indirection, kv_cache_block_pointers)
        context = self.dense(context)
        context = RaggedTensor.from_row_lengths(context, hidden_states.row_lengths, hidden_states.max_row_length)
        return (context, present) if use_cache else context
