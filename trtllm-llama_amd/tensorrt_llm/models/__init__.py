from .llama.model import LLaMADecoderLayer, LLaMAForCausalLM, LLaMAModel
from .quantized.quant import smooth_quantize, weight_only_quantize

__all__ = ['LLaMADecoderLayer', 'LLaMAForCausalLM', 'LLaMAModel', 'smooth_quantize', 'weight_only_quantize']
