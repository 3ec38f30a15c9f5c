"""Linear / ColumnLinear / RowLinear (T/tensorrt_llm/layers/linear.py): weight stored [out, in], y = x W^T through the
Gemm plugin when plugin_config.gemm_plugin is set (transb=True), TP split on dim 0 (column) or dim 1 (row)."""
from .._common import default_net
from ..functional import allgather, allreduce, gemm_plugin, matmul
from ..module import Module
from ..parameter import Parameter


def _gemm_plugin(input, mat2, transa=False, transb=False):
    return gemm_plugin(input, mat2, transa, transb)


class Linear(Module):

    def __init__(self, in_features, out_features, bias=True, dtype=None, tp_group=None, tp_size=1, gather_output=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features // tp_size
        self.dtype = dtype
        self.weight = Parameter(shape=(self.out_features, self.in_features), dtype=dtype)
        self.tp_size = tp_size
        self.tp_group = tp_group
        self.gather_output = gather_output
        if bias:
            raise NotImplementedError('bias is not built: LLaMA linear layers have none (llama_model.py:60,73)')
        self.register_parameter('bias', None)

    def forward(self, x):
        if default_net().plugin_config.gemm_plugin:
            x = _gemm_plugin(x, self.weight.value, transb=True)
        else:
            x = matmul(x, self.weight.value, transb=True)
        if self.gather_output and self.tp_size > 1 and self.tp_group is not None:
            # [B, V/tp] per rank -> all-gather along dim 0 -> [tp * B, V/tp]; the engine re-interleaves to [B, V]
            x = allgather(x, self.tp_group)
        return x


ColumnLinear = Linear


class RowLinear(Module):

    def __init__(self, in_features, out_features, bias=True, dtype=None, tp_group=None, tp_size=1):
        super().__init__()
        self.in_features = in_features // tp_size
        self.out_features = out_features
        self.dtype = dtype
        self.weight = Parameter(shape=(self.out_features, self.in_features), dtype=dtype)
        if bias:
            raise NotImplementedError('bias is not built: LLaMA linear layers have none')
        self.register_parameter('bias', None)
        self.tp_group = tp_group
        self.tp_size = tp_size

    def forward(self, x):
        if default_net().plugin_config.gemm_plugin:
            x = _gemm_plugin(x, self.weight.value, transb=True)
        else:
            x = matmul(x, self.weight.value, transb=True)
        if self.tp_size > 1 and self.tp_group is not None:
            x = allreduce(x, self.tp_group)
        return x
