"""FT-checkpoint writer for the LLaMA int8 paths: int8 KV cache and SmoothQuant (T/examples/llama_quant/convert.py).

On-disk format kept from the reference (it is what weight.py::load_from_ft_llama and the reference's weight_quant.py
read): `model.<key>[.<rank>].bin`, raw little-endian arrays, weights stored [in, out] (QKV [in, 3, out])."""
import numpy as np


def split(v, tp_size, idx, dim=0):
    if tp_size == 1:
        return v
    if len(v.shape) == 1:
        return np.ascontiguousarray(np.split(v, tp_size)[idx])
    return np.ascontiguousarray(np.split(v, tp_size, axis=dim)[idx])


def save_val(val, dir, key, tp_num=None):
    suffix = 'bin' if tp_num is None else f'{tp_num}.bin'
    np.ascontiguousarray(val).tofile(dir / f'model.{key}.{suffix}')


def save_split(split_vals, dir, key, i, factor):
    for j, val in enumerate(split_vals):
        save_val(val, dir, key, i * factor + j)


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, 'detach') else np.asarray(x)


def generate_int8(weights, act_range, is_qkv=False, multi_query_mode=False):
    """Quantised weights (per tensor and per output column) and the scaling factors of one GEMM (convert.py:27-103):
      scale_x_orig_quant (T)     127 / max|x|                      activation -> int8, before the GEMM
      scale_y_quant_orig (T)     max|y| / 127                      int8 GEMM output -> fp  (also the int8 KV-cache scale)
      scale_w_quant_orig (T, C)  max|w| / 127                      used with per-token dynamic activation scales
      scale_y_accum_quant (T, C) (127/max|y|) / ((127/max|x|)(127/max|w|))   int32 accumulator -> int8 range
    QKV: `weights` is [in, 3, out]; "per tensor" means one factor for each of Q, K, V."""
    w_absmax = _np(act_range['w']).astype(np.float32)
    if is_qkv and multi_query_mode:
        raise ValueError('Multi-query w/ int8 quant has not been supported yet')
    if is_qkv:
        scale_w_orig_quant_t = 127. / w_absmax.reshape(3, -1).max(axis=-1, keepdims=True)
        scale_w_orig_quant_c = 127. / w_absmax.reshape(3, -1)
    else:
        scale_w_orig_quant_t = 127. / w_absmax.max()
        scale_w_orig_quant_c = 127. / w_absmax
    scale_w_quant_orig_t = 1.0 / scale_w_orig_quant_t
    scale_w_quant_orig_c = 1.0 / scale_w_orig_quant_c
    x_max, y_max = float(_np(act_range['x']).max()), float(_np(act_range['y']).max())
    scale_x_orig_quant_t = np.array(127. / x_max)
    scale_y_orig_quant_t = np.array(127. / y_max)
    scale_y_quant_orig_t = np.array(y_max / 127.)
    scale_y_accum_quant_t = scale_y_orig_quant_t / (scale_x_orig_quant_t * scale_w_orig_quant_t)
    scale_y_accum_quant_c = scale_y_orig_quant_t / (scale_x_orig_quant_t * scale_w_orig_quant_c)
    if is_qkv:
        scale_y_accum_quant_t = np.broadcast_to(scale_y_accum_quant_t, scale_w_orig_quant_c.shape)
        scale_w_quant_orig_t = np.broadcast_to(scale_w_quant_orig_t, scale_w_orig_quant_c.shape)
    if hasattr(weights, 'detach'):
        # a torch tensor (possibly on the GPU: a 7B model is 6.5e9 weights, minutes in numpy): the same arithmetic -
        # float32 product, round-half-even, clip to +-127 - on the tensor's own device; the int8 results stay there
        import torch
        wt = weights.detach().to(torch.float32)
        to_i8 = lambda s: (wt * torch.from_numpy(np.ascontiguousarray(s, dtype=np.float32)).to(wt.device)).round_() \
            .clamp_(-127, 127).to(torch.int8)
    else:
        weights = np.asarray(weights)
        to_i8 = lambda s: (weights * s).round().clip(-127, 127).astype(np.int8)
    return {
        'weight.int8': to_i8(scale_w_orig_quant_t),
        'weight.int8.col': to_i8(scale_w_orig_quant_c),
        'scale_x_orig_quant': scale_x_orig_quant_t.astype(np.float32),
        'scale_w_quant_orig': np.asarray(scale_w_quant_orig_t).astype(np.float32),
        'scale_w_quant_orig.col': scale_w_quant_orig_c.astype(np.float32),
        'scale_y_accum_quant': np.asarray(scale_y_accum_quant_t).astype(np.float32),
        'scale_y_accum_quant.col': scale_y_accum_quant_c.astype(np.float32),
        'scale_y_quant_orig': scale_y_quant_orig_t.astype(np.float32),
    }


def write_int8(vals, dir, base_key, split_dim, tp_rank, split_factor, kv_cache_only=False):
    """convert.py:106-146: int8 weights split like their fp counterparts; per-tensor factors once; per-column factors
    split with the columns for column-parallel GEMMs (split_dim == -1), whole for row-parallel ones."""
    if not kv_cache_only:
        save_split(np.split(vals['weight.int8'], split_factor, axis=split_dim), dir, f'{base_key}.weight.int8', tp_rank,
                   split_factor)
        save_split(np.split(vals['weight.int8.col'], split_factor, axis=split_dim), dir, f'{base_key}.weight.int8.col',
                   tp_rank, split_factor)
    saved_keys_once = ['scale_y_quant_orig']
    if not kv_cache_only:
        saved_keys_once += ['scale_x_orig_quant', 'scale_w_quant_orig', 'scale_y_accum_quant']
        if split_dim == -1:
            for k in ('scale_w_quant_orig.col', 'scale_y_accum_quant.col'):
                save_split(np.split(vals[k], split_factor, axis=split_dim), dir, f'{base_key}.{k}', tp_rank, split_factor)
        else:
            saved_keys_once += ['scale_w_quant_orig.col', 'scale_y_accum_quant.col']
    if tp_rank == 0:
        for k in saved_keys_once:
            save_val(vals[k], dir, f'{base_key}.{k}')


def split_and_save_weight(tp_rank, saved_dir, split_factor, key, vals, storage_type, act_range, config):
    """One tensor of the HF model -> its FT files (convert.py:160-325).  `vals`: numpy array, linear weights already
    transposed to [in, out] (QKV [in, 3, out]); `storage_type` a numpy dtype; config: {'int8_outputs': None |
    'kv_cache_only' | 'all', 'multi_query_mode': bool}."""
    int8_outputs = config.get('int8_outputs', None)
    multi_query_mode = config.get('multi_query_mode', False)
    save_int8 = int8_outputs in ('all', 'kv_cache_only')
    vals = np.asarray(vals)
    fp = vals.astype(storage_type)
    if 'layernorm.weight' in key or 'layernorm.bias' in key:
        if tp_rank == 0:
            save_val(fp, saved_dir, key)
    elif 'attention.dense.weight' in key or 'mlp.down_proj.weight' in key:
        save_split(np.split(fp, split_factor, axis=0), saved_dir, key, tp_rank, split_factor)  # row parallel
        if act_range is not None and int8_outputs == 'all':
            write_int8(generate_int8(vals, act_range, multi_query_mode=multi_query_mode), saved_dir, key.replace('.weight', ''),
                       0, tp_rank, split_factor)
    elif 'mlp.gate_proj.weight' in key or 'mlp.up_proj.weight' in key:
        save_split(np.split(fp, split_factor, axis=-1), saved_dir, key, tp_rank, split_factor)  # column parallel
        if act_range is not None and int8_outputs == 'all':
            write_int8(generate_int8(vals, act_range, multi_query_mode=multi_query_mode), saved_dir, key.replace('.weight', ''),
                       -1, tp_rank, split_factor)
    elif 'attention.query_key_value.weight' in key:
        save_val(fp, saved_dir, key)  # kept whole: the loader splits the heads (weight.py::split_qkv)
        if save_int8:
            write_int8(generate_int8(vals, act_range, is_qkv=True, multi_query_mode=False), saved_dir,
                       key.replace('.weight', ''), -1, tp_rank, split_factor, kv_cache_only=int8_outputs == 'kv_cache_only')
    else:
        print(f'[WARNING] {key} not handled by converter')
