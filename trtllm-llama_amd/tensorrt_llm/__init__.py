"""tensorrt_llm-shaped front-end of the MI355X LLaMA decoder path: same module / layer / builder / runtime surface as
the reference package (T/tensorrt_llm/__init__.py) for everything the llama_quant example touches; the engine it
builds runs on hand-written gfx950 kernels behind the C ABI of include/*.h (no TensorRT)."""
from . import functional, profiler
from ._common import default_net, net_guard, precision
from ._utils import mpi_rank, mpi_world_size, str_dtype_to_np, str_dtype_to_torch, str_dtype_to_trt
from .builder import Builder, BuilderConfig
from .functional import RaggedTensor, Tensor
from .logger import logger
from .mapping import Mapping
from .module import Module, ModuleList
from .network import Network
from .parameter import Parameter
from .plugin import _load_plugin_lib

__version__ = '0.1.3'  # T/setup.py:22

__all__ = ['logger', 'str_dtype_to_trt', 'str_dtype_to_np', 'str_dtype_to_torch', 'mpi_rank', 'mpi_world_size',
           'default_net', 'net_guard', 'precision', 'Network', 'Mapping', 'Builder', 'BuilderConfig', 'Tensor',
           'RaggedTensor', 'Parameter', 'Module', 'ModuleList', 'functional', 'profiler', '_load_plugin_lib']
