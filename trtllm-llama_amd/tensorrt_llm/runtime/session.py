"""`Session` (T/tensorrt_llm/runtime/session.py:37-160): the reference's generic engine runner - deserialise an engine, query
its I/O, run it on a stream.  Kept because user code imports it (`from tensorrt_llm.runtime import Session`,
`Session.from_serialized_engine`); there is no TensorRT here, so the engine is the traced LLaMA network of builder.py and the
one thing a Session can run is that network's context phase through the C++ session (csrc/runtime/session.cpp) -
GenerationSession is the decode loop."""
from dataclasses import dataclass
from typing import Any, Dict, List

import numpy as np

from .native import NativeSession


@dataclass
class TensorInfo:
    name: str
    dtype: Any
    shape: tuple


class Session(object):

    def __init__(self, **kwargs):
        # use Session.from_serialized_engine to create a session (as in the reference)
        self._native = None
        self._engine = None

    def _init(self, engine_buffer):
        self._engine = bytes(engine_buffer)
        self._native = NativeSession(engine=self._engine)  # parses + verifies the traced network, uploads the weights
        self._shape = None
        return self

    @staticmethod
    def from_serialized_engine(engine) -> 'Session':
        return Session()._init(engine)

    @property
    def engine(self) -> bytes:
        return self._engine

    @property
    def native(self) -> NativeSession:
        return self._native

    @property
    def context(self):
        return self._native

    def infer_shapes(self, inputs: List[TensorInfo], context=None) -> List[TensorInfo]:
        """Takes the shape of `input_ids` ([batch, max_input_len]) and answers with the logits' shape."""
        ids = [i for i in inputs if i.name == 'input_ids']
        if not ids or len(ids[0].shape) != 2:
            return None
        self._shape = tuple(int(d) for d in ids[0].shape)
        return [TensorInfo('logits', np.float32, (self._shape[0], -1))]

    def run(self, inputs: Dict[str, Any], outputs: Dict[str, Any], stream=0, context=None) -> bool:
        """Context phase of the engine: inputs `input_ids` [batch, len] and `input_lengths` [batch] (numpy or torch, host or
        device), output `logits` [batch, vocab] fp32 written into outputs['logits'] (numpy array or torch tensor)."""
        def host(a):
            return a.detach().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a)
        try:
            ids, lens = host(inputs['input_ids']).astype(np.int32), host(inputs['input_lengths']).astype(np.int32)
            out = outputs['logits']
            vocab = int(out.shape[-1])
            if (self._native.batch, self._native.max_in) != ids.shape:
                self._native.setup(ids.shape[0], ids.shape[1], 1)
            self._native.context(ids, lens, int(stream) if stream else 0)
            logits = self._native.logits(vocab, int(stream) if stream else 0)
            if hasattr(out, 'copy_'):
                import torch
                out.copy_(torch.from_numpy(logits).to(out.device))
            else:
                out[...] = logits
            return True
        except (RuntimeError, KeyError):
            return False
