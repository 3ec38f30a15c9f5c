"""Parameter: a named weight of a Module (T/tensorrt_llm/parameter.py).  `.value` is a numpy array; reading the
value of a parameter that was never assigned materialises the reference's default init (Xavier-uniform for
matrices, ones for vectors: parameter.py:28-38)."""
import math
from typing import Optional, Sequence, Union

import numpy as np

from ._utils import DataType, str_dtype_to_trt, trt_dtype_to_np


class Parameter(object):
    _DEFAULT_DTYPE = DataType.FLOAT

    def __init__(self, value: Optional[np.ndarray] = None, shape: Sequence[int] = None,
                 dtype: Union[str, DataType, None] = None):
        if dtype is None:
            dtype = Parameter._DEFAULT_DTYPE
        if isinstance(dtype, str):
            dtype = str_dtype_to_trt(dtype)
        self._dtype = DataType(dtype)
        self._shape = tuple(shape) if shape is not None else (tuple(value.shape) if value is not None else None)
        self._value = None
        if value is not None:
            self.value = value

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return self._dtype

    def _default(self):
        np_dt = trt_dtype_to_np(self._dtype)
        shape = self._shape
        if len(shape) == 2 and np.issubdtype(np_dt, np.floating):
            # Xavier uniform, fan_in + fan_out over the two dims (parameter.py:28-38)
            v_range = math.sqrt(6) / math.sqrt(shape[0] + shape[1])
            rng = np.random.default_rng(abs(hash(shape)) % (2**32))
            return (rng.uniform(-v_range, v_range, shape)).astype(np_dt)
        if np.issubdtype(np_dt, np.floating):
            return np.ones(shape, dtype=np_dt)
        return np.zeros(shape, dtype=np_dt)

    @property
    def raw_value(self) -> np.ndarray:
        """The numpy array (materialising the default init if nothing was assigned)."""
        if self._value is None:
            self._value = self._default()
        return self._value

    @property
    def value(self):
        """Inside `net_guard(network)`: the constant tensor of this parameter in the traced graph (what the layers'
        forward() consume, like the reference's trt constant); outside: the numpy array."""
        from ._common import default_net, has_default_net
        if not has_default_net():
            return self.raw_value
        net = default_net()
        cached = getattr(self, '_tensor', None)
        if cached is not None and cached[0] is net:
            return cached[1]
        from .functional import constant
        t = constant(None, parameter=self)
        self._tensor = (net, t)
        return t

    @value.setter
    def value(self, v):
        v = np.ascontiguousarray(v)
        if self._shape is not None and tuple(v.shape) != tuple(self._shape):
            # the reference asserts shape equality; accept the same number of BYTES so that processed weight-only
            # layouts ([N, ldw] int8) can be assigned to the fp32-typed [K, N/4] parameters the loaders expect
            want = int(np.prod(self._shape)) * self._dtype.itemsize
            # (rows of the processed layout are padded to 16 / 32 elements when K is not a multiple: >= then)
            smuggled = v.dtype in (np.int8, np.uint8) and self._dtype == DataType.FLOAT
            assert v.nbytes == want or (smuggled and v.nbytes > want), \
                f'Parameter shape mismatch: expected {self._shape} {self._dtype.name}, got {v.shape} {v.dtype}'
        self._value = v

    def is_inited(self):
        return self._value is not None
