"""`Session` (T/tensorrt_llm/runtime/session.py:37-160): the reference's generic engine runner - deserialise an engine, query
its I/O, run it on a stream.  Kept because user code imports it (`from tensorrt_llm.runtime import Session`,
`Session.from_serialized_engine`); there is no TensorRT here, so the engine is the traced LLaMA network of builder.py and the
one thing a Session can run is that network's context phase through the C++ session (csrc/runtime/session.cpp) -
GenerationSession is the decode loop."""
from dataclasses import dataclass
from typing import Any, Dict, List

import numpy as np

from ..logger import logger
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
        """Takes the shape of `input_ids` ([batch, max_input_len]) and answers with the logits' shape, [batch, vocab_size] - the
        real vocabulary size of the engine.  As in the reference (runtime/session.py:116-145) a tensor that is not an input of
        the engine, or one of the wrong dtype, is an error: logged, and the answer is None."""
        known = {'input_ids': np.int32, 'input_lengths': np.int32}
        for i in inputs:
            if i.name not in known:
                logger.error(f'Tensor:{i.name} is not an input tensor')
                return None
            if i.dtype is not None and np.dtype(_np_dtype(i.dtype)) != np.dtype(known[i.name]):
                logger.error(f'Tensor:{i.name} has wrong dtype')
                return None
        ids = [i for i in inputs if i.name == 'input_ids']
        if not ids or len(ids[0].shape) != 2 or min(int(d) for d in ids[0].shape) < 1:
            logger.error('Tensor:input_ids must be given with a [batch, max_input_len] shape')
            return None
        self._shape = tuple(int(d) for d in ids[0].shape)
        return [TensorInfo('logits', np.float32, (self._shape[0], int(self._native.vocab or -1)))]

    def run(self, inputs: Dict[str, Any], outputs: Dict[str, Any], stream=0, context=None) -> bool:
        """Context phase of the engine: inputs `input_ids` [batch, len] and `input_lengths` [batch] (numpy or torch, host or
        device), output `logits` [batch, vocab] fp32 written into outputs['logits'] (numpy array or torch tensor).  Returns False
        - with the cause logged - for missing tensors, shapes that do not fit each other or the engine, and launch failures
        (the reference's run() returns the enqueue's verdict, runtime/session.py:147-190); it does not raise on bad input."""

        def host(a):
            return a.detach().cpu().numpy() if hasattr(a, 'detach') else np.asarray(a)
        try:
            for name in ('input_ids', 'input_lengths'):
                if name not in inputs:
                    raise ValueError(f'missing input tensor {name}')
            if 'logits' not in outputs:
                raise ValueError('missing output tensor logits')
            ids, lens = host(inputs['input_ids']), host(inputs['input_lengths'])
            if ids.ndim != 2 or ids.shape[0] < 1 or ids.shape[1] < 1:
                raise ValueError(f'input_ids must be [batch, max_input_len], got shape {tuple(ids.shape)}')
            if lens.ndim != 1 or lens.shape[0] != ids.shape[0]:
                raise ValueError(f'input_lengths must be [batch = {ids.shape[0]}], got shape {tuple(lens.shape)}')
            if not (np.issubdtype(ids.dtype, np.integer) and np.issubdtype(lens.dtype, np.integer)):
                raise ValueError('input_ids / input_lengths must be integer tensors')
            ids, lens = ids.astype(np.int32), lens.astype(np.int32)
            if lens.min() < 1 or lens.max() > ids.shape[1]:
                raise ValueError(f'input_lengths must lie in [1, {ids.shape[1]}]')
            out = outputs['logits']
            vocab = int(self._native.vocab or out.shape[-1])
            if tuple(int(d) for d in out.shape) != (ids.shape[0], vocab):
                raise ValueError(f'logits must be [{ids.shape[0]}, {vocab}], got shape {tuple(out.shape)}')
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
        except (RuntimeError, KeyError, ValueError, AssertionError, IndexError, TypeError) as e:
            logger.error(f'Session.run failed: {e}')
            return False


def _np_dtype(dt):
    """numpy dtype of a TensorInfo.dtype given as numpy dtype / type, torch dtype or the reference's string names."""
    if isinstance(dt, str):
        return {'int32': np.int32, 'float32': np.float32, 'float16': np.float16}.get(dt, dt)
    name = str(dt)
    if name.startswith('torch.'):
        return {'torch.int32': np.int32, 'torch.float32': np.float32, 'torch.float16': np.float16, 'torch.int64': np.int64}.get(name, np.void)
    return dt
