from .generation import GenerationSession, ModelConfig, SamplingConfig

__all__ = ['GenerationSession', 'ModelConfig', 'SamplingConfig']
