"""Weight loaders of the llama_quant example.

load_from_hf_llama  — T/examples/llama/weight.py:29-177: HF LlamaForCausalLM (or a {name: ndarray} state dict)
                      -> Parameter.value, TP split rules :86-172 (QKV reshape(3, D, D) split on the head dim,
                      o_proj / down_proj split on the input dim, gate/up on the output dim, lm_head on the vocab),
                      weight-only quantisation through the library's symmetric_quantize_last_axis (:101-110).
load_from_ft_llama  — T/examples/llama_quant/weight_quant.py:84-446: the FT directory written by hf_llama_convert.py
                      (file set in SURVEY.md Appendix B), incl. SmoothQuant int8 weights + scales and int8-KV scales.
                      The reference loader is TP-broken for QKV (reads [D, 3D/tp] out of one unsplit file,
                      :244-247); this one splits per q/k/v head block like weight.py.
"""
import configparser
import time
from pathlib import Path

import numpy as np

import tensorrt_llm
from tensorrt_llm.parameter import Parameter
from tensorrt_llm._utils import str_dtype_to_np
from tensorrt_llm.plugin import capi
from tensorrt_llm.quantization import QuantMode


def _np(x):
    if hasattr(x, 'detach'):
        x = x.detach().float().cpu().numpy()
    return np.asarray(x)


def split(v, tp_size, idx, dim=0):
    if tp_size == 1:
        return v
    if v.ndim == 1:
        return np.ascontiguousarray(np.split(v, tp_size)[idx])
    return np.ascontiguousarray(np.split(v, tp_size, axis=dim)[idx])


def split_qkv(qkv_3d_d, tp_size, idx):
    """[3*D, D] -> reshape(3, D, D) -> split dim 1 (heads) -> [3*D/tp, D] (weight.py:86-100)."""
    three_d, d = qkv_3d_d.shape
    w = qkv_3d_d.reshape(3, three_d // 3, d)
    return np.ascontiguousarray(split(w, tp_size, idx, dim=1).reshape(-1, d))


def _set_linear(module, w_out_in, quant_mode, np_dtype):
    """fp16 [out, in] weight -> module parameters for the module's quantisation mode."""
    if quant_mode.is_weight_only():
        bits = 8 if quant_mode.is_int8_weight_only() else 4
        processed, scales, _ = capi.symmetric_quantize_last_axis(np.ascontiguousarray(w_out_in.T.astype(np.float16)), bits)
        module.weight.value = processed  # bytes of the fp32 [in, out/4|8] view
        module.per_channel_scale.value = scales.astype(np_dtype)
    else:
        module.weight.value = np.ascontiguousarray(w_out_in.astype(np_dtype))


def load_from_hf_llama(tensorrt_llm_llama, hf_llama, rank=0, tensor_parallel=1, dtype='float16', multi_query_mode=False,
                       kv_scales=None):
    tensorrt_llm.logger.info('Loading weights from HF LLaMA...')
    tik = time.time()
    assert not multi_query_mode, 'multi_query_mode is not built'
    quant_mode = getattr(tensorrt_llm_llama, 'quant_mode', QuantMode(0))
    assert not quant_mode.has_act_and_weight_quant(), 'SmoothQuant engines load from the FT directory (calibrated scales)'
    np_dtype = str_dtype_to_np(dtype)
    sd = dict(hf_llama.state_dict()) if hasattr(hf_llama, 'state_dict') else dict(hf_llama)
    get = lambda k: _np(sd[k])
    m = tensorrt_llm_llama
    m.vocab_embedding.weight.value = get('model.embed_tokens.weight').astype(np_dtype)
    m.ln_f.weight.value = get('model.norm.weight').astype(np_dtype)
    head = get('lm_head.weight')
    vpad = -head.shape[0] % tensor_parallel  # vocab padded to a multiple of tp (T/tensorrt_llm/_utils.py:194-195)
    if vpad:
        head = np.pad(head, ((0, vpad), (0, 0)))
    m.lm_head.weight.value = np.ascontiguousarray(split(head, tensor_parallel, rank).astype(np_dtype))
    for i, layer in enumerate(m.layers):
        p = f'model.layers.{i}.'
        qkv = np.concatenate([get(p + f'self_attn.{n}_proj.weight') for n in 'qkv'], axis=0)
        layer.input_layernorm.weight.value = get(p + 'input_layernorm.weight').astype(np_dtype)
        layer.post_layernorm.weight.value = get(p + 'post_attention_layernorm.weight').astype(np_dtype)
        _set_linear(layer.attention.qkv, split_qkv(qkv, tensor_parallel, rank), quant_mode, np_dtype)
        _set_linear(layer.attention.dense, split(get(p + 'self_attn.o_proj.weight'), tensor_parallel, rank, dim=1),
                    quant_mode, np_dtype)
        # naming trap: fc <-> gate_proj, gate <-> up_proj, proj <-> down_proj (T/tests/test_layer.py:158-160)
        _set_linear(layer.mlp.fc, split(get(p + 'mlp.gate_proj.weight'), tensor_parallel, rank, dim=0), quant_mode, np_dtype)
        _set_linear(layer.mlp.gate, split(get(p + 'mlp.up_proj.weight'), tensor_parallel, rank, dim=0), quant_mode, np_dtype)
        _set_linear(layer.mlp.proj, split(get(p + 'mlp.down_proj.weight'), tensor_parallel, rank, dim=1), quant_mode, np_dtype)
        if quant_mode.has_int8_kv_cache():
            assert kv_scales is not None, 'int8 KV cache needs calibrated scales: kv_scales[layer] = max|qkv| / 127'
            s = np.float32(kv_scales[i])
            layer.attention.kv_orig_quant_scale.value = np.array([1.0 / s], np.float32)
            layer.attention.kv_quant_orig_scale.value = np.array([s], np.float32)
    tensorrt_llm.logger.info(f'Weights loaded. Total time: {time.time() - tik:.1f} s')


def parse_ft_config(ini_file):
    cfg = configparser.ConfigParser()
    cfg.read(ini_file)
    s = cfg['llama']
    n_embd = s.getint('hidden_size')
    n_head = s.getint('num_attention_heads')
    n_layer = s.getint('num_hidden_layers')
    n_positions = s.getint('max_position_embeddings')
    vocab_size = s.getint('vocab_size')
    hidden_act = s.get('hidden_act', 'silu')
    inter_size = s.getint('intermediate_size', fallback=None)
    multi_query_mode = s.getboolean('multi_query_mode', fallback=False)
    dtype = s.get('storage_dtype', 'float16')
    return n_embd, n_head, n_layer, n_positions, vocab_size, True, hidden_act, 1.0, False, inter_size, multi_query_mode, dtype, 0, 0


def load_from_ft_llama(tensorrt_llm_llama, dir_path, rank=0, tensor_parallel=1, dtype='float16'):
    tensorrt_llm.logger.info('Loading weights from FT LLaMA...')
    tik = time.time()
    quant_mode = getattr(tensorrt_llm_llama, 'quant_mode', QuantMode(0))
    n_embd, n_head, n_layer, _, vocab_size, _, _, _, _, inter_size, *_ = parse_ft_config(Path(dir_path) / 'config.ini')
    np_dtype = str_dtype_to_np(dtype)
    d = Path(dir_path)
    m = tensorrt_llm_llama
    sq = quant_mode.has_act_and_weight_quant()
    per_ch, per_tok = quant_mode.has_per_channel_scaling(), quant_mode.has_per_token_dynamic_scaling()

    def fromfile(name, shape=None, dt=None):
        p = d / name
        if not p.exists():
            return None
        t = np.fromfile(p, dtype=np_dtype if dt is None else dt)
        return t.reshape(shape) if shape is not None else t

    def need(name, shape=None, dt=None):
        t = fromfile(name, shape, dt)
        assert t is not None, f'{d / name} is missing'
        return t

    m.vocab_embedding.weight.value = need('model.wte.weight.bin', [vocab_size, n_embd])
    m.ln_f.weight.value = need('model.final_layernorm.weight.bin')
    head = need('model.lm_head.weight.bin', [vocab_size, n_embd])
    vpad = -vocab_size % tensor_parallel
    if vpad:
        head = np.pad(head, ((0, vpad), (0, 0)))
    m.lm_head.weight.value = np.ascontiguousarray(split(head, tensor_parallel, rank))

    def set_sq(module, base, out_full, in_full, kind):
        """kind: 'qkv' | 'col' (output split) | 'row' (input split).  Files hold [in, out] int8 (convert.py::write_int8);
        QKV is [in, 3, out / tp] per rank (reference converter) or [in, 3, out] whole (older directories)."""
        suffix = 'int8.col' if per_ch else 'int8'
        tp = tensor_parallel
        if kind == 'qkv':
            w = fromfile(f'{base}.weight.{suffix}.{rank}.bin', [in_full, 3, out_full // 3 // tp], np.int8)
            if w is not None:
                w = np.ascontiguousarray(w.transpose(1, 2, 0).reshape(out_full // tp, in_full))  # [3*out/tp, in]
            else:
                w = need(f'{base}.weight.{suffix}.bin', [in_full, 3, out_full // 3], np.int8)
                w = split_qkv(np.ascontiguousarray(w.transpose(1, 2, 0).reshape(out_full, in_full)), tp, rank)
        elif kind == 'col':
            w = need(f'{base}.weight.{suffix}.{rank}.bin', [in_full, out_full // tp], np.int8).T
        else:
            w = need(f'{base}.weight.{suffix}.{rank}.bin', [in_full // tp, out_full], np.int8).T
        module.weight.value = np.ascontiguousarray(w)
        key = 'scale_w_quant_orig' if per_tok else 'scale_y_accum_quant'
        if kind == 'qkv':
            # one factor per output channel, or ("per tensor") one for each of Q, K and V stored broadcast to [3, out]:
            # either way the plugin gets a per-channel vector, [1, 3 * out / tp] in (q, k, v) order
            s = fromfile(f'{base}.{key}.col.{rank}.bin', None, np.float32) if per_ch else None
            if s is None:
                s = need(f'{base}.{key}.col.bin' if per_ch else f'{base}.{key}.bin', None, np.float32)
                s = split(s.reshape(3, -1), tp, rank, dim=1)
            s = s.reshape(1, -1)
            if s.shape[1] != out_full // tp:  # a genuinely scalar file
                s = np.full((1, out_full // tp), s.reshape(-1)[0], np.float32)
            if tuple(module.per_channel_scale.shape) != s.shape:
                module.per_channel_scale = Parameter(shape=s.shape, dtype='float32')
        elif per_ch:
            if kind == 'col':
                s = need(f'{base}.{key}.col.{rank}.bin', None, np.float32).reshape(1, -1)
            else:
                s = need(f'{base}.{key}.col.bin', None, np.float32).reshape(1, -1)
        else:
            s = need(f'{base}.{key}.bin', None, np.float32).reshape(-1)[:1].reshape(1, 1)
        module.per_channel_scale.value = np.ascontiguousarray(s.astype(np.float32))
        if not per_tok:
            module.act_scale.value = need(f'{base}.scale_y_quant_orig.bin', None, np.float32).reshape(-1)[:1].reshape(1, 1)

    for i, layer in enumerate(m.layers):
        p = f'model.model.layers.{i}.'
        layer.input_layernorm.weight.value = need(p + 'input_layernorm.weight.bin')
        layer.post_layernorm.weight.value = need(p + 'post_attention_layernorm.weight.bin')
        if sq:
            set_sq(layer.attention.qkv, p + 'attention.query_key_value', 3 * n_embd, n_embd, 'qkv')
            set_sq(layer.attention.dense, p + 'attention.dense', n_embd, n_embd, 'row')
            set_sq(layer.mlp.fc, p + 'mlp.gate_proj', inter_size, n_embd, 'col')
            set_sq(layer.mlp.gate, p + 'mlp.up_proj', inter_size, n_embd, 'col')
            set_sq(layer.mlp.proj, p + 'mlp.down_proj', n_embd, inter_size, 'row')
            if not per_tok:
                f1 = lambda n: need(n, None, np.float32).reshape(-1)[:1]
                layer.input_layernorm.scale_to_int.value = f1(p + 'attention.query_key_value.scale_x_orig_quant.bin')
                layer.attention.quantization_scaling_factor.value = f1(p + 'attention.dense.scale_x_orig_quant.bin')
                layer.post_layernorm.scale_to_int.value = f1(p + 'mlp.gate_proj.scale_x_orig_quant.bin')
                layer.mlp.quantization_scaling_factor.value = f1(p + 'mlp.down_proj.scale_x_orig_quant.bin')
        else:
            qkv = need(p + 'attention.query_key_value.weight.bin', [n_embd, 3, n_embd])  # [in, 3, out]
            qkv = np.ascontiguousarray(qkv.transpose(1, 2, 0).reshape(3 * n_embd, n_embd))
            _set_linear(layer.attention.qkv, split_qkv(qkv, tensor_parallel, rank), quant_mode, np_dtype)
            _set_linear(layer.attention.dense,
                        need(p + f'attention.dense.weight.{rank}.bin', [n_embd // tensor_parallel, n_embd]).T, quant_mode, np_dtype)
            _set_linear(layer.mlp.fc, need(p + f'mlp.gate_proj.weight.{rank}.bin', [n_embd, inter_size // tensor_parallel]).T,
                        quant_mode, np_dtype)
            _set_linear(layer.mlp.gate, need(p + f'mlp.up_proj.weight.{rank}.bin', [n_embd, inter_size // tensor_parallel]).T,
                        quant_mode, np_dtype)
            _set_linear(layer.mlp.proj, need(p + f'mlp.down_proj.weight.{rank}.bin', [inter_size // tensor_parallel, n_embd]).T,
                        quant_mode, np_dtype)
        if quant_mode.has_int8_kv_cache():
            # kv_quant_orig = scale_y_quant_orig of the QKV output; kv_orig_quant = 1 / that (weight_quant.py:439-446)
            t = need(p + 'attention.query_key_value.scale_y_quant_orig.bin', None, np.float32).reshape(-1)[:1]
            layer.attention.kv_orig_quant_scale.value = (1.0 / t).astype(np.float32)
            layer.attention.kv_quant_orig_scale.value = t.astype(np.float32)
    tensorrt_llm.logger.info(f'Weights loaded. Total time: {time.time() - tik:.1f} s')
