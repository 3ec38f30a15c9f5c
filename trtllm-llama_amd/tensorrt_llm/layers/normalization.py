"""RmsNorm (T/tensorrt_llm/layers/normalization.py:33-54): eps 1e-6, optional elementwise weight."""
from ..functional import rms_norm
from ..module import Module
from ..parameter import Parameter


class RmsNorm(Module):

    def __init__(self, normalized_shape, eps=1e-06, elementwise_affine=True, dtype=None):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape, )
        self.normalized_shape = tuple(normalized_shape)
        self.elementwise_affine = elementwise_affine
        if elementwise_affine:
            self.weight = Parameter(shape=self.normalized_shape, dtype=dtype)
        else:
            self.register_parameter('weight', None)
        self.eps = eps

    def forward(self, x):
        weight = None if self.weight is None else self.weight.value
        return rms_norm(x, self.normalized_shape, weight, self.eps)
