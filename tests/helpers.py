"""Test-side helpers: call the plugin C-ABI with torch tensors (torch only owns the device memory)."""
import numpy as np
import torch

from tensorrt_llm.plugin import capi

_T2C = {torch.float32: capi.FLOAT, torch.float16: capi.HALF, torch.int8: capi.INT8, torch.int32: capi.INT32,
        torch.uint8: capi.UINT8}


def dtype_code(t: torch.Tensor) -> int:
    return _T2C[t.dtype]


def make_plugin(name, fields):
    p = capi.Plugin.create(name, [capi.PluginField(k, v) for k, v in fields])
    assert p is not None, f'{name}: {capi.last_error()}'
    return p


def i32(v):
    return np.array(v, dtype=np.int32)


def i8(v):
    return np.array(v, dtype=np.int8)


def f32(v):
    return np.array(v, dtype=np.float32)


class HostTensor:
    """A CPU int32 tensor passed by host pointer (GPTAttention input 3) or a shape-only tensor."""

    def __init__(self, arr, shape=None):
        self.arr = np.ascontiguousarray(arr, dtype=np.int32)
        self.shape = list(self.arr.shape) if shape is None else list(shape)

    def ptr(self):
        return self.arr.ctypes.data


def run_plugin(plugin, inputs, outputs, override_in_shapes=None, override_in_types=None):
    """inputs/outputs: torch cuda tensors or HostTensor.  Enqueues on torch's current stream and synchronises."""
    in_shapes, in_types, in_ptrs = [], [], []
    for i, t in enumerate(inputs):
        if isinstance(t, HostTensor):
            in_shapes.append(t.shape)
            in_types.append(capi.INT32)
            in_ptrs.append(t.ptr())
        else:
            in_shapes.append(list(t.shape))
            in_types.append(dtype_code(t))
            in_ptrs.append(t.data_ptr())
    if override_in_shapes:
        for k, v in override_in_shapes.items():
            in_shapes[k] = list(v)
    if override_in_types:
        for k, v in override_in_types.items():
            in_types[k] = v
    out_shapes = [list(t.shape) for t in outputs]
    out_types = [dtype_code(t) for t in outputs]
    out_ptrs = [t.data_ptr() for t in outputs]
    ws_bytes = plugin.workspace_size(in_shapes, in_types, out_shapes, out_types)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    plugin.enqueue(in_shapes, in_types, in_ptrs, out_shapes, out_types, out_ptrs, ws.data_ptr(), stream)
    torch.cuda.synchronize()


def h(x):
    """numpy float -> torch fp16 cuda"""
    return torch.from_numpy(np.asarray(x, dtype=np.float32)).to(torch.float16).cuda()


def as_f32(t: torch.Tensor) -> np.ndarray:
    return t.detach().float().cpu().numpy()
