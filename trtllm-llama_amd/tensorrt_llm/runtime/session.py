"""Session (T/tensorrt_llm/runtime/session.py): thin owner of a deserialised engine."""
from .native import NativeSession


class Session(object):

    def __init__(self, **kwargs):
        self._native = None

    @staticmethod
    def from_serialized_engine(engine: bytes) -> 'Session':
        s = Session()
        s._native = NativeSession(engine=engine)
        return s

    @property
    def native(self) -> NativeSession:
        return self._native
