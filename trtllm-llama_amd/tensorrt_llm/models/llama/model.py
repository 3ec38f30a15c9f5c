"""LLaMA model definition with the constructor / forward signatures of the reference
(T/examples/llama_quant/llama_model.py:23-35,78-86,122-137,159-168,209-222,253-263,289-290 — identical to
T/tensorrt_llm/models/llama/model.py plus the `quant_mode` argument) and the attribute names the weight loaders
address: input_layernorm, attention(.qkv, .dense, .kv_orig_quant_scale, .kv_quant_orig_scale),
mlp(.fc, .gate, .proj), post_layernorm, vocab_embedding, ln_f, lm_head."""
from collections import OrderedDict

from ..._common import default_net
from ..._utils import DataType, pad_vocab_size, str_dtype_to_trt
from ...functional import RaggedTensor, Tensor, assertion, expand_mask, gather_last_token_logits, shape
from ...layers import (Attention, AttentionMaskType, ColumnLinear, Embedding, GatedMLP, PositionEmbeddingType, RmsNorm)
from ...module import Module, ModuleList
from ...quantization import QuantMode


class LLaMADecoderLayer(Module):
    """h = x + O(attn(QKV(rms(x)))) ; y = h + proj(silu(fc(rms(h))) * gate(rms(h)))."""

    def __init__(self, layer_id, hidden_size, num_attention_heads, max_position_embeddings, dtype=None,
                 hidden_act='silu', mlp_hidden_size=None, neox_rotary_style=True, multi_query_mode=False,
                 tp_group=None, tp_size=1, quant_mode=QuantMode(0)):
        super().__init__()
        self._layer_id = layer_id
        # kept on the layer because smooth_quantize() rebuilds the sub-modules from them
        self.hidden_size = hidden_size
        self.num_attention_heads = num_attention_heads
        self.max_position_embeddings = max_position_embeddings
        self.dtype = dtype
        self.hidden_act = hidden_act
        self.mlp_hidden_size = mlp_hidden_size if mlp_hidden_size else hidden_size * 4
        self.neox_rotary_style = neox_rotary_style
        self.multi_query_mode = multi_query_mode
        self.attention_mask_type = AttentionMaskType.causal
        self.position_embedding_type = PositionEmbeddingType.rope
        self.tp_group = tp_group
        self.tp_size = tp_size
        self.quant_mode = quant_mode

        self.input_layernorm = RmsNorm(normalized_shape=hidden_size, dtype=dtype)
        self.attention = Attention(hidden_size, num_attention_heads, max_position_embeddings, dtype=dtype,
                                   attention_mask_type=AttentionMaskType.causal, bias=False,
                                   position_embedding_type=PositionEmbeddingType.rope,
                                   neox_rotary_style=neox_rotary_style, multi_query_mode=multi_query_mode,
                                   tp_group=tp_group, tp_size=tp_size,
                                   use_int8_kv_cache=quant_mode.has_int8_kv_cache())
        self.mlp = GatedMLP(hidden_size=hidden_size, ffn_hidden_size=self.mlp_hidden_size, hidden_act=hidden_act,
                            dtype=dtype, bias=False, tp_group=tp_group, tp_size=tp_size)
        self.post_layernorm = RmsNorm(normalized_shape=hidden_size, dtype=dtype)

    def forward(self, hidden_states: RaggedTensor, attention_mask=None, past_key_value=None, sequence_length=None,
                past_key_value_length=None, masked_tokens=None, use_cache=False, cache_indirection=None,
                kv_cache_block_pointers=None):
        x = hidden_states.data
        lengths, max_len = hidden_states.row_lengths, hidden_states.max_row_length
        normed = RaggedTensor.from_row_lengths(self.input_layernorm(x), lengths, max_len)
        attn = self.attention(normed, attention_mask=attention_mask, past_key_value=past_key_value,
                              sequence_length=sequence_length, past_key_value_length=past_key_value_length,
                              masked_tokens=masked_tokens, use_cache=use_cache, cache_indirection=cache_indirection,
                              kv_cache_block_pointers=kv_cache_block_pointers)
        presents = None
        if use_cache:
            attn, presents = attn
        h = x + attn.data
        out = h + self.mlp(self.post_layernorm(h))
        out = RaggedTensor.from_row_lengths(out, attn.row_lengths, attn.max_row_length)
        return (out, presents) if use_cache else out


class LLaMAModel(Module):

    def __init__(self, num_layers, num_heads, hidden_size, vocab_size, hidden_act, max_position_embeddings, dtype,
                 mlp_hidden_size=None, neox_rotary_style=True, tensor_parallel=1, tensor_parallel_group=None,
                 multi_query_mode=False, quant_mode=QuantMode(0)):
        super().__init__()
        self.vocab_embedding = Embedding(vocab_size, hidden_size, dtype=dtype)  # replicated under TP
        self.layers = ModuleList([
            LLaMADecoderLayer(layer_id=i, hidden_size=hidden_size, num_attention_heads=num_heads,
                              max_position_embeddings=max_position_embeddings, dtype=dtype, hidden_act=hidden_act,
                              mlp_hidden_size=mlp_hidden_size, neox_rotary_style=neox_rotary_style,
                              multi_query_mode=multi_query_mode, tp_group=tensor_parallel_group,
                              tp_size=tensor_parallel, quant_mode=quant_mode) for i in range(num_layers)
        ])
        self.ln_f = RmsNorm(normalized_shape=hidden_size, dtype=dtype)

    def forward(self, input_ids: RaggedTensor, position_ids=None, past_key_value=None, sequence_length=None,
                past_key_value_length=None, masked_tokens=None, use_cache=False, attention_mask=None,
                cache_indirection=None, kv_cache_block_pointers=None):
        hidden = self.vocab_embedding(input_ids.data)
        if past_key_value is None:
            past_key_value = tuple([None] * len(self.layers))
        if kv_cache_block_pointers is None:
            kv_cache_block_pointers = tuple([None] * len(self.layers))
        if attention_mask is not None:
            attention_mask = expand_mask(attention_mask, shape(input_ids.data, -1))
        hidden = RaggedTensor.from_row_lengths(hidden, input_ids.row_lengths, input_ids.max_row_length)
        presents = []
        for layer, past, pointers in zip(self.layers, past_key_value, kv_cache_block_pointers):
            hidden = layer(hidden, past_key_value=past, sequence_length=sequence_length,
                           past_key_value_length=past_key_value_length, masked_tokens=masked_tokens,
                           use_cache=use_cache, attention_mask=attention_mask, cache_indirection=cache_indirection,
                           kv_cache_block_pointers=pointers)
            if use_cache:
                hidden, present = hidden
                presents.append(present)
        hidden = self.ln_f(hidden.data)
        return (hidden, tuple(presents)) if use_cache else hidden


class LLaMAForCausalLM(LLaMAModel):

    def __init__(self, num_layers, num_heads, hidden_size, vocab_size, hidden_act, max_position_embeddings, dtype,
                 mlp_hidden_size=None, neox_rotary_style=True, tensor_parallel=1, tensor_parallel_group=None,
                 multi_query_mode=False, quant_mode=QuantMode(0)):
        self.kv_dtype = str_dtype_to_trt(dtype) if isinstance(dtype, str) else DataType(dtype)
        if quant_mode.has_int8_kv_cache():
            self.kv_dtype = str_dtype_to_trt('int8')
        self.quant_mode = quant_mode
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.hidden_size = hidden_size
        self.vocab_size = vocab_size
        self.tensor_parallel = tensor_parallel
        self._multi_query_mode = multi_query_mode
        super().__init__(num_layers, num_heads, hidden_size, vocab_size, hidden_act, max_position_embeddings, dtype,
                         mlp_hidden_size, neox_rotary_style, tensor_parallel, tensor_parallel_group,
                         multi_query_mode, quant_mode)
        self.inter_size = mlp_hidden_size if mlp_hidden_size else hidden_size * 4
        vocab_size_padded = pad_vocab_size(vocab_size, tensor_parallel)
        self.lm_head = ColumnLinear(hidden_size, vocab_size_padded, bias=False, dtype=dtype,
                                    tp_group=tensor_parallel_group, tp_size=tensor_parallel, gather_output=True)

    def forward(self, input_ids: RaggedTensor, position_ids=None, past_key_value=None, sequence_length=None,
                past_key_value_length=None, masked_tokens=None, use_cache=False, last_token_ids=None,
                attention_mask=None, cache_indirection=None, kv_cache_block_pointers=None):
        hidden = super().forward(input_ids, position_ids, past_key_value, sequence_length, past_key_value_length,
                                 masked_tokens, use_cache, attention_mask, cache_indirection, kv_cache_block_pointers)
        presents = None
        if use_cache:
            hidden, presents = hidden
        hidden = gather_last_token_logits(hidden, last_token_ids, default_net().plugin_config.remove_input_padding)
        lm_logits = self.lm_head(hidden)  # [batch, hidden] -> [batch, vocab]
        lm_logits.mark_output('logits', str_dtype_to_trt('float32'))  # llama_model.py:279
        if use_cache:
            for i, present in enumerate(presents):
                present.mark_output(f'present_key_value_{i}', self.kv_dtype)
            return lm_logits, presents
        return lm_logits

    def prepare_inputs(self, max_batch_size, max_input_len, max_new_tokens, use_cache, max_beam_width):
        """Symbolic network inputs with the reference's tensor names (llama_model.py:289-435; generation.py:188-208)."""
        head_size = self.hidden_size // self.num_heads
        num_heads = self.num_heads // self.tensor_parallel
        max_len = max_input_len + max_new_tokens
        bb = [1, (max_batch_size * max_beam_width + 1) // 2, max_batch_size * max_beam_width]
        bs = [1, (max_batch_size + 1) // 2, max_batch_size]
        beams = [1, (max_beam_width + 1) // 2, max_beam_width]
        inlen = [1, 1, max_input_len]
        lens = [0, (max_len + 1) // 2, max_len]
        remove_input_padding = default_net().plugin_config.remove_input_padding
        if not default_net().plugin_config.gpt_attention_plugin:
            raise ValueError('the LLaMA path needs plugin_config.set_gpt_attention_plugin() (RoPE lives in the plugin)')
        i32 = DataType.INT32

        def named(name, shape, ranges):
            return Tensor(name=name, dtype=i32, shape=shape, dim_range=OrderedDict(ranges))

        if remove_input_padding:
            # the real tokens of the whole batch back to back (llama_model.py:312-331)
            ntok = [1, (max_input_len * max_batch_size + 1) // 2, max_input_len * max_batch_size]
            input_ids = named('input_ids', [1, -1], [('batch_size_fake', [1]), ('num_tokens', [ntok])])
            position_ids = named('position_ids', [1, -1], [('batch_size_fake', [1]), ('num_tokens', [ntok])])
        else:
            input_ids = named('input_ids', [-1, -1], [('batch_size', [bb]), ('input_len', [inlen])])
            position_ids = named('position_ids', [-1, -1], [('batch_size', [bb]), ('input_len', [inlen])])
        past_key_value = []
        paged = default_net().plugin_config.paged_kv_cache
        block_pointers = []
        for i in range(self.num_layers):
            if paged:
                # the block pool [blocks, 2, heads, tokens_per_block, head_size] and, per layer, the table of block pointers
                # int64 [batch, beam, 2, max_blocks] carried as int32 pairs (T/tensorrt_llm/models/gpt/model.py prepare_inputs)
                tpb = default_net().plugin_config.tokens_per_block
                nblk = [1, (max_len // tpb + 2) // 2, -(-max_len // tpb)]
                kv = Tensor(name=f'past_key_value_{i}', dtype=self.kv_dtype, shape=[-1, 2, num_heads, tpb, head_size],
                            dim_range=OrderedDict([('blocks', [[1, bb[2] * nblk[2] // 2 + 1, bb[2] * nblk[2]]]), ('kv', [2]),
                                                   ('num_heads', [num_heads]), ('tokens_per_block', [tpb]),
                                                   ('head_size', [head_size])]))
                block_pointers.append(named(f'kv_cache_block_pointers_{i}', [-1, -1, 2, -1],
                                            [('batch_size', [bs]), ('beam_width', [beams]), ('kv', [2]),
                                             ('max_blocks_x2', [[2 * n for n in nblk]])]))
            else:
                kv = Tensor(name=f'past_key_value_{i}', dtype=self.kv_dtype, shape=[-1, 2, num_heads, -1, head_size],
                            dim_range=OrderedDict([('batch_size', [bb]), ('kv', [2]), ('num_heads', [num_heads]),
                                                   ('past_key_len', [lens]), ('head_size', [head_size])]))
            past_key_value.append(kv)
            assertion(shape(input_ids, 0), 'batch size')
        sequence_length = named('sequence_length', [-1], [('batch_size', [bb])])
        past_key_value_length = named('past_key_value_length', [-1], [('past_key_value_length', [lens])])
        masked_tokens = named('masked_tokens', [-1, -1], [('batch_size', [bb]), ('max_seq_len', [lens])])
        input_lengths = named('input_lengths', [-1], [('batch_size', [bb])])
        max_input_length = named('max_input_length', [-1], [('max_input_len', [inlen])])
        last_token_ids = named('last_token_ids', [-1], [('batch_size', [bb])])
        cache_indirection = named('cache_indirection', [-1, -1, -1],
                                  [('batch_size', [bs]), ('beam_width', [beams]), ('max_seq_len', [lens])])
        input_ids_ragged = RaggedTensor.from_row_lengths(input_ids, input_lengths, max_input_length)
        inputs = (input_ids_ragged, position_ids, past_key_value, sequence_length, past_key_value_length, masked_tokens,
                  True, last_token_ids, None, cache_indirection)
        return inputs + (block_pointers, ) if paged else inputs
