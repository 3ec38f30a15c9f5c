"""Plugin library loader + PluginConfig (same flags/setters as T/tensorrt_llm/plugin/plugin.py:33-140; the flags are
serialised into config.json["plugin_config"] by Builder.save_config and read back by run.py)."""
from enum import IntEnum

from . import capi

_TRT_LLM_PLUGIN_NAMESPACE = capi.TRT_LLM_PLUGIN_NAMESPACE


def _load_plugin_lib():
    """ctypes.CDLL(<pkg>/libs/libnvinfer_plugin_tensorrt_llm.so, RTLD_GLOBAL) + initLibNvInferPlugins."""
    capi.load_library()


class ContextFMHAType(IntEnum):
    disabled = 0
    enabled = 1  # fp16 accumulation in the reference's FMHA cubins
    enabled_with_fp32_acc = 2


class PluginConfig(object):

    def __init__(self) -> None:
        self.init()

    def init(self):
        self.bert_attention_plugin = False
        self.gpt_attention_plugin = False
        self.inflight_batching_gpt_attention_plugin = False
        self.identity_plugin = False
        self.gemm_plugin = False
        self.smooth_quant_gemm_plugin = False
        self.layernorm_plugin = False
        self.layernorm_quantization_plugin = False
        self.attention_qk_half_accumulation = False
        self.remove_input_padding = False
        self.context_fmha_type = ContextFMHAType.disabled
        self.weight_only_quant_matmul_plugin = False
        self.nccl_plugin = False
        self.quantize_per_token_plugin = False
        self.quantize_tensor_plugin = False
        self.paged_kv_cache = False
        self.tokens_per_block = 64
        self.lookup_plugin = False
        self.in_flight_batching = False
        # MI355X addition: the RMSNorm(+int8 quant) plugin that LLaMA's SmoothQuant needs (SURVEY "fact 1")
        self.rmsnorm_quantization_plugin = False

    def _set(self, name, value, note):
        from ..logger import logger
        setattr(self, name, value)
        logger.info(note)
        return self

    def enable_qk_half_accum(self):
        return self._set('attention_qk_half_accumulation', True, 'Attention BMM1(QK) accumulation type is set to FP16')

    def set_context_fmha(self, context_fmha_type=ContextFMHAType.enabled):
        assert isinstance(context_fmha_type, ContextFMHAType)
        return self._set('context_fmha_type', context_fmha_type, f'Context FMHA {context_fmha_type.name}')

    def enable_remove_input_padding(self):
        return self._set('remove_input_padding', True, 'Remove Padding Enabled')

    def enable_paged_kv_cache(self):
        return self._set('paged_kv_cache', True, 'Paged KV Cache Enabled')

    def enable_in_flight_batching(self):
        return self._set('in_flight_batching', True, 'In-flight Batching Enabled')

    def set_gpt_attention_plugin(self, dtype='float16'):
        return self._set('gpt_attention_plugin', dtype, f'GPT Attention plugin: {dtype}')

    def set_inflight_batching_gpt_attention_plugin(self, dtype='float16'):
        return self._set('inflight_batching_gpt_attention_plugin', dtype, f'IB GPT Attention plugin: {dtype}')

    def set_bert_attention_plugin(self, dtype='float16'):
        return self._set('bert_attention_plugin', dtype, f'BERT Attention plugin: {dtype}')

    def set_identity_plugin(self, dtype='float16'):
        return self._set('identity_plugin', dtype, f'Identity plugin: {dtype}')

    def set_gemm_plugin(self, dtype='float16'):
        return self._set('gemm_plugin', dtype, f'GEMM plugin: {dtype}')

    def set_smooth_quant_gemm_plugin(self, dtype='float16'):
        return self._set('smooth_quant_gemm_plugin', dtype, f'SmoothQuant GEMM plugin: {dtype}')

    def set_layernorm_plugin(self, dtype='float16'):
        return self._set('layernorm_plugin', dtype, f'LayerNorm plugin: {dtype}')

    def set_layernorm_quantization_plugin(self, dtype='float16'):
        return self._set('layernorm_quantization_plugin', dtype, f'LayerNorm quantization plugin: {dtype}')

    def set_rmsnorm_quantization_plugin(self, dtype='float16'):
        return self._set('rmsnorm_quantization_plugin', dtype, f'RMSNorm quantization plugin: {dtype}')

    def set_weight_only_quant_matmul_plugin(self, dtype='float16'):
        return self._set('weight_only_quant_matmul_plugin', dtype, f'Weight-only quant matmul plugin: {dtype}')

    def set_nccl_plugin(self, dtype='float16'):
        return self._set('nccl_plugin', dtype, f'NCCL (RCCL) plugin: {dtype}')

    def set_quantize_per_token_plugin(self):
        return self._set('quantize_per_token_plugin', True, 'Quantize per token plugin enabled')

    def set_quantize_tensor_plugin(self):
        return self._set('quantize_tensor_plugin', True, 'Quantize tensor plugin enabled')

    def set_lookup_plugin(self, dtype='float16'):
        return self._set('lookup_plugin', dtype, f'Lookup plugin: {dtype}')
