"""Network: the traced graph (T/tensorrt_llm/network.py).  TensorRT is replaced by a plain node list:
each functional op appends {op, inputs, outputs, attrs}; Builder.build_engine serialises it next to the weights."""
import collections

from .plugin import PluginConfig


class Network(object):

    def __init__(self, **kwargs):
        self._plugin_config = PluginConfig()
        self._inputs = collections.OrderedDict()
        self._outputs = collections.OrderedDict()
        self._nodes = []
        self._named_parameters = None
        self._constants = []  # (tensor name, Parameter)
        self._counter = 0

    @property
    def plugin_config(self):
        return self._plugin_config

    @property
    def trt_network(self):
        return self

    def new_name(self, hint='t'):
        self._counter += 1
        return f'{hint}_{self._counter}'

    def add_input(self, tensor):
        self._inputs[tensor.name] = tensor

    def add_node(self, op, inputs, outputs, **attrs):
        self._nodes.append(dict(op=op, inputs=[t.name for t in inputs], outputs=[t.name for t in outputs], attrs=attrs))
        return self._nodes[-1]

    def mark_output(self, tensor, name, dtype):
        self._outputs[name] = (tensor, dtype)
        self._nodes.append(dict(op='mark_output', inputs=[tensor.name], outputs=[name], attrs=dict(dtype=int(dtype))))

    def set_named_parameters(self, named_parameters):
        self._named_parameters = list(named_parameters)

    @property
    def named_parameters(self):
        return self._named_parameters

    def get_inputs(self):
        return list(self._inputs.values())

    def get_outputs(self):
        return [t for t, _ in self._outputs.values()]

    @property
    def nodes(self):
        return self._nodes

    def ops(self):
        return [n['op'] for n in self._nodes]
from ._common import net_guard  # noqa: F401  (the examples import it from here: build.py)
