from .mode import QuantMode

__all__ = ['QuantMode']
