"""MLP / GatedMLP (T/tensorrt_llm/layers/mlp.py).  GatedMLP.forward = proj(act(fc(x)) * gate(x)) with the naming the
loaders rely on: fc <-> HF gate_proj, gate <-> HF up_proj, proj <-> HF down_proj (T/tests/test_layer.py:158-160)."""
from ..functional import ACT2FN
from ..module import Module
from .linear import ColumnLinear, RowLinear


class MLP(Module):

    def __init__(self, hidden_size, ffn_hidden_size, hidden_act, bias=True, dtype=None, tp_group=None, tp_size=1):
        super().__init__()
        if hidden_act not in ACT2FN:
            raise ValueError(f'unsupported activation function: {hidden_act}')
        self.fc = ColumnLinear(hidden_size, ffn_hidden_size, bias=bias, dtype=dtype, tp_group=tp_group, tp_size=tp_size,
                               gather_output=False)
        self.proj = RowLinear(ffn_hidden_size, hidden_size, bias=bias, dtype=dtype, tp_group=tp_group, tp_size=tp_size)
        self.hidden_act = hidden_act
        self.dtype = dtype

    def forward(self, hidden_states):
        return self.proj(ACT2FN[self.hidden_act](self.fc(hidden_states)))


class GatedMLP(MLP):

    def __init__(self, hidden_size, ffn_hidden_size, hidden_act, bias=True, dtype=None, tp_group=None, tp_size=1):
        super().__init__(hidden_size, ffn_hidden_size, hidden_act, bias=bias, dtype=dtype, tp_group=tp_group,
                         tp_size=tp_size)
        self.gate = ColumnLinear(hidden_size, ffn_hidden_size, bias=bias, dtype=dtype, tp_group=tp_group,
                                 tp_size=tp_size, gather_output=False)

    def forward(self, hidden_states):
        inter = ACT2FN[self.hidden_act](self.fc(hidden_states))
        return self.proj(inter * self.gate(hidden_states))
