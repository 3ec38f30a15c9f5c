from .generation import GenerationSession, ModelConfig, SamplingConfig
from .session import Session

__all__ = ['GenerationSession', 'ModelConfig', 'SamplingConfig', 'Session']
