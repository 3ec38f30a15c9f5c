"""Builder (T/tensorrt_llm/builder.py:57-267) without TensorRT: `build_engine(network, builder_config)` serialises the
traced network + the weights into a single-file engine ("TLLMENG1": config text, tensor table, data) that the C++
host loop (include/tllm_runtime_api.h) loads; `save_config` writes the same config.json
({"builder_config": {...}, "plugin_config": {...}}, builder.py:259-267) that run.py / summarize.py read back."""
import json
import os
import struct
import time
from pathlib import Path
from typing import Union

import numpy as np

from ._common import _is_building
from ._utils import DataType, np_dtype_to_trt, str_dtype_to_trt
from .logger import logger
from .network import Network


class BuilderConfig(object):

    def __init__(self, **kwargs):
        self._values = dict(kwargs)
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self):
        return dict(self._values)


class Builder():
    _ALLOWED_PRECISIONS = ['float32', 'float16']

    def __init__(self):
        super().__init__()
        self.strongly_typed = False

    def create_network(self) -> Network:
        return Network()

    def create_builder_config(self, precision: str, timing_cache: Union[str, Path, None] = None,
                              tensor_parallel: int = 1, use_refit: bool = False, int8: bool = False,
                              opt_level: int = None, **kwargs) -> BuilderConfig:
        """kwargs become fields of config.json["builder_config"] (name, num_layers, num_heads, hidden_size, vocab_size,
        hidden_act, max_position_embeddings, max_batch_size, max_input_len, max_output_len, multi_query_mode, ...)."""
        if precision == 'bfloat16':
            raise ValueError('bfloat16 engines are not built on MI355X (fp16 storage / fp32 accumulate)')
        assert precision in self._ALLOWED_PRECISIONS, f'precision should be one of {self._ALLOWED_PRECISIONS}'
        return BuilderConfig(precision=precision, tensor_parallel=tensor_parallel, use_refit=use_refit, int8=int8,
                             opt_level=opt_level, **kwargs)

    @_is_building
    def build_engine(self, network: Network, builder_config: BuilderConfig) -> bytes:
        """Returns the serialized engine (bytes), or None on failure like the reference."""
        assert isinstance(network, Network)
        t0 = time.time()
        named = network.named_parameters
        if named is None:
            logger.error('build_engine: call network.set_named_parameters(model.named_parameters()) first')
            return None
        cfg = builder_config.to_dict()
        required = ('num_layers', 'num_heads', 'hidden_size', 'vocab_size')
        for k in required:
            if k not in cfg:
                logger.error(f'build_engine: builder_config lacks {k}')
                return None
        ops = network.ops()
        plugins = [n['attrs']['plugin_type'] for n in network.nodes if n['op'] == 'plugin']
        if plugins.count('GPTAttention') != cfg['num_layers']:
            logger.error('build_engine: the traced network is not a LLaMA stack '
                         f"({plugins.count('GPTAttention')} GPTAttention nodes for {cfg['num_layers']} layers)")
            return None
        if 'logits' not in network._outputs:
            logger.error("build_engine: the network has no 'logits' output")
            return None
        tensors = []
        inter_size = cfg.get('inter_size')
        for name, p in named:
            v = p.raw_value
            if name.endswith('mlp.fc.weight') and inter_size is None and v.dtype == np.float16:
                inter_size = int(v.shape[0]) * int(cfg.get('tensor_parallel', 1))
            tensors.append((name, np.ascontiguousarray(v)))
        if inter_size is None:
            logger.error('build_engine: pass inter_size=... to create_builder_config for quantised engines')
            return None
        header = {
            'num_layers': cfg['num_layers'], 'num_heads': cfg['num_heads'], 'hidden_size': cfg['hidden_size'],
            'inter_size': inter_size, 'vocab_size': cfg['vocab_size'],
            'max_position_embeddings': cfg.get('max_position_embeddings', 2048),
            'rms_norm_eps': cfg.get('rms_norm_eps', 1e-6), 'tp_size': cfg.get('tensor_parallel', 1),
            'tp_rank': cfg.get('tp_rank', 0), 'quant_mode': int(cfg.get('quant_mode', 0)),
            'neox_rotary_style': 1, 'precision': cfg['precision'],
            'remove_input_padding': 1 if network.plugin_config.remove_input_padding else 0,
            'paged_kv_cache': 1 if network.plugin_config.paged_kv_cache else 0,
            'tokens_per_block': int(getattr(network.plugin_config, 'tokens_per_block', 64)),
            'network_ops': ','.join(ops[:0]),  # the node list itself goes below as one JSON line
        }
        tactics = self._profile_gemm_tactics(cfg, inter_size)
        if tactics:
            header['gemm_tactics'] = tactics
        text = '\n'.join(f'{k}={v}' for k, v in header.items())
        # which Parameter (module path) every constant tensor of the graph is: lets the runtime check that each plugin port
        # is fed by the weight its schedule reads there
        path_of = {id(p): name for name, p in named}
        constants = {tname: path_of.get(id(v), '') for tname, v in network._constants}
        text += '\nnetwork_json=' + json.dumps(dict(inputs=[t.name for t in network.get_inputs()],
                                                    outputs=list(network._outputs.keys()), constants=constants,
                                                    nodes=[dict(op=n['op'], inputs=n['inputs'], outputs=n['outputs'],
                                                                attrs={k: v for k, v in n['attrs'].items()})
                                                           for n in network.nodes if n['op'] != 'constant']))
        blob = bytearray()
        blob += b'TLLMENG1'
        tb = text.encode()
        blob += struct.pack('<Q', len(tb)) + tb
        blob += struct.pack('<Q', len(tensors))
        off = 0
        table = []
        for name, v in tensors:
            nb = name.encode()
            dt = int(np_dtype_to_trt(v.dtype)) if v.dtype != np.uint8 else int(DataType.INT8)
            dims = list(v.shape) if v.ndim > 0 else [1]
            blob += struct.pack('<I', len(nb)) + nb + struct.pack('<ii', dt, len(dims))
            blob += struct.pack(f'<{len(dims)}q', *dims) + struct.pack('<QQ', v.nbytes, off)
            table.append((off, v))
            off += (v.nbytes + 63) // 64 * 64
        pad = (-len(blob)) % 64
        blob += b'\0' * pad
        data0 = len(blob)
        blob += b'\0' * off
        for o, v in table:
            blob[data0 + o:data0 + o + v.nbytes] = v.tobytes()
        logger.info(f'Build engine time: {time.time() - t0:.2f} s, {len(blob) / 2**20:.1f} MiB')
        return bytes(blob)

    @staticmethod
    def _profile_gemm_tactics(cfg: dict, inter_size: int) -> str:
        """The reference's SmoothQuant GEMM plugin profiles its tile configurations on the device for every M bucket while the
        engine is built and keeps the winners in the engine (int8_gemm_template.h:372-457, smoothQuantGemmPlugin.cpp:253-282).
        Same here when a GPU is visible at build time: the four GEMM shapes of a layer at every power-of-two M up to
        max_batch_size * max_input_len (and at that M itself) are timed by `tllm_gemm_profile`; the table travels in the engine
        header (`gemm_tactics=`).  Without a GPU (or with TLLM_GEMM_TACTICS=off) the session profiles at set-up instead."""
        if os.environ.get('TLLM_GEMM_TACTICS', '').lower() in ('off', '0'):
            return ''
        try:
            import ctypes

            import torch
            if not torch.cuda.is_available():
                return ''
            from .plugin import capi
            lib = capi.load_library()
        except Exception:
            return ''
        lib.tllm_gemm_profile.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        lib.tllm_gemm_tactics_export.argtypes = [ctypes.c_char_p, ctypes.c_int64]
        lib.tllm_gemm_tactics_export.restype = ctypes.c_int64
        tp = int(cfg.get('tensor_parallel', 1))
        D, inter = int(cfg['hidden_size']), int(inter_size)
        m_max = int(cfg.get('max_batch_size', 1)) * int(cfg.get('max_input_len', 0))
        if m_max < 32:
            return ''
        qm = int(cfg.get('quant_mode', 0))
        if (qm & 3) and not (qm & 4):  # weight-only (no activation quantisation)
            # weight-only prefill dequantises in the GEMM's main loop (csrc/kernels/gemm_woq.hip): one kernel per shape class,
            # the fp16 tactic table is never consulted - nothing to profile
            return ''
        wtype = 3 if qm & 4 else 0  # SmoothQuant int8, else fp16
        shapes = {(3 * D // tp, D), (D, D // tp), (inter // tp, D), (D, inter // tp)}
        ms = sorted({m_max} | {1 << i for i in range(5, 14) if (1 << i) < m_max})
        for m in ms:
            for n, k in shapes:
                if lib.tllm_gemm_profile(wtype, m, n, k, None, None, None):
                    logger.warning(f'gemm tactic profile failed for {m} x {n} x {k}: {capi.last_error()}')
                    return ''
        need = lib.tllm_gemm_tactics_export(None, 0)
        buf = ctypes.create_string_buffer(int(need))
        lib.tllm_gemm_tactics_export(buf, need)
        return buf.value.decode()

    @staticmethod
    def save_timing_cache(builder_config: BuilderConfig, out_path: str) -> bool:
        """TensorRT's timing cache has one counterpart here, the prefill GEMM tactic table (`_profile_gemm_tactics`), and that
        travels inside the engine; nothing to save separately.  Kept for build.py compatibility."""
        return True

    @staticmethod
    def save_config(builder_config: BuilderConfig, config_path: str):
        cfg = builder_config.to_dict()
        plugin_config = cfg.pop('plugin_config', None)
        out = {'builder_config': {k: (v if isinstance(v, (int, float, str, bool, type(None), list, dict)) else str(v))
                                  for k, v in cfg.items()}}
        if plugin_config is not None:
            out['plugin_config'] = {k: (v if isinstance(v, (int, float, str, bool, type(None))) else int(v))
                                    for k, v in plugin_config.__dict__.items()}
        with open(config_path, 'w') as f:
            json.dump(out, f, indent=4)
        logger.info(f'Config saved to {config_path}.')
