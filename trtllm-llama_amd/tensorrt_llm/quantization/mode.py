"""QuantMode — the quantisation flag word that travels from build.py into every layer.

Same bit layout and predicate names as the reference (T/tensorrt_llm/quantization/mode.py:4-137):
INT4_WEIGHTS=1, INT8_WEIGHTS=2, ACTIVATIONS=4, PER_CHANNEL=8, PER_TOKEN=16, INT8_KV_CACHE=32,
FP8_KV_CACHE=64, COUNT=128.  Pinned by tests/golden/quant_mode.json (truth table generated from the reference).
"""
from enum import IntFlag


class QuantMode(IntFlag):
    INT4_WEIGHTS = 1
    INT8_WEIGHTS = 2
    ACTIVATIONS = 4
    PER_CHANNEL = 8
    PER_TOKEN = 16
    INT8_KV_CACHE = 32
    FP8_KV_CACHE = 64
    COUNT = 128

    WEIGHTS_AND_ACTIVATIONS = 1 | 2 | 4
    VALID_FLAGS = 127

    # -- bit tests ------------------------------------------------------------------------------------------
    def _all(self, bits, mask=127):
        return (int(self) & int(mask)) == int(bits)

    def _any(self, bits):
        return (int(self) & int(bits)) != 0

    def is_int8_weight_only(self):
        return self._all(QuantMode.INT8_WEIGHTS, QuantMode.WEIGHTS_AND_ACTIVATIONS)

    def is_int4_weight_only(self):
        return self._all(QuantMode.INT4_WEIGHTS, QuantMode.WEIGHTS_AND_ACTIVATIONS)

    def is_weight_only(self):
        return self.is_int4_weight_only() or self.is_int8_weight_only()

    def has_act_and_weight_quant(self):
        return self._all(QuantMode.INT8_WEIGHTS | QuantMode.ACTIVATIONS, QuantMode.WEIGHTS_AND_ACTIVATIONS)

    def has_per_token_dynamic_scaling(self):
        return self._any(QuantMode.PER_TOKEN)

    def has_act_static_scaling(self):
        return not self.has_per_token_dynamic_scaling()

    def has_per_channel_scaling(self):
        return self._any(QuantMode.PER_CHANNEL)

    def has_int8_kv_cache(self):
        return self._any(QuantMode.INT8_KV_CACHE)

    def has_fp8_kv_cache(self):
        return self._any(QuantMode.FP8_KV_CACHE)

    def has_any_quant(self):
        return self._any(QuantMode.INT8_WEIGHTS | QuantMode.ACTIVATIONS | QuantMode.INT8_KV_CACHE
                         | QuantMode.FP8_KV_CACHE)

    def set_int8_kv_cache(self):
        return self | QuantMode.INT8_KV_CACHE

    def set_fp8_kv_cache(self):
        return self | QuantMode.FP8_KV_CACHE

    # -- constructors ---------------------------------------------------------------------------------------
    @staticmethod
    def from_description(quantize_weights=False, quantize_activations=False, per_token=False, per_channel=False,
                         use_int4_weights=False, use_int8_kv_cache=False, use_fp8_kv_cache=False):
        bad = (quantize_activations and not quantize_weights) or \
              ((per_token or per_channel) and not (quantize_weights and quantize_activations))
        if bad:
            raise ValueError('Unsupported combination of QuantMode args: '
                             f'{quantize_weights=}, {quantize_activations=}, {per_token=}, {per_channel=}, '
                             f'{use_int4_weights=}, {use_int8_kv_cache=}, {use_fp8_kv_cache=}')
        bits = 0
        if quantize_weights:
            bits |= QuantMode.INT4_WEIGHTS if use_int4_weights else QuantMode.INT8_WEIGHTS
        for on, flag in ((quantize_activations, QuantMode.ACTIVATIONS), (per_channel, QuantMode.PER_CHANNEL),
                         (per_token, QuantMode.PER_TOKEN), (use_int8_kv_cache, QuantMode.INT8_KV_CACHE),
                         (use_fp8_kv_cache, QuantMode.FP8_KV_CACHE)):
            if on:
                bits |= flag
        return QuantMode(bits)

    @staticmethod
    def use_smooth_quant(per_token=False, per_channel=False):
        return QuantMode.from_description(True, True, per_token, per_channel)

    @staticmethod
    def use_weight_only(use_int4_weights=False):
        return QuantMode.from_description(True, False, False, False, use_int4_weights)
