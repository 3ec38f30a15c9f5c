from .generation import GenerationSession, ModelConfig, SamplingConfig
from .kv_cache_manager import GenerationSequence, KVCacheManager
from .session import Session, TensorInfo

__all__ = ['ModelConfig', 'GenerationSession', 'GenerationSequence', 'KVCacheManager', 'SamplingConfig', 'Session',
           'TensorInfo']
