"""ctypes binding of include/tllm_plugin_api.h — the Python half of the drop-in boundary.

Replaces what the reference reaches through TensorRT's Python bindings
(trt.get_plugin_registry().get_plugin_creator(...).create_plugin(...), IExecutionContext enqueue;
T/tensorrt_llm/plugin/plugin.py:7-22, T/tensorrt_llm/functional.py:2826-2893).
Fails loudly when the HIP library is missing: there is no CPU fallback.
"""
import ctypes
import os
from typing import List, Optional, Sequence

import numpy as np

TRT_LLM_PLUGIN_NAMESPACE = 'tensorrt_llm'
_LIB_NAME = 'libnvinfer_plugin_tensorrt_llm.so'

# nvinfer1::DataType
FLOAT, HALF, INT8, INT32, BOOL, UINT8, FP8 = 0, 1, 2, 3, 4, 5, 6
# nvinfer1::PluginFieldType
PF_FLOAT16, PF_FLOAT32, PF_FLOAT64, PF_INT8, PF_INT16, PF_INT32, PF_CHAR, PF_DIMS, PF_UNKNOWN = range(9)

MAX_DIMS = 8


class Dims(ctypes.Structure):
    _fields_ = [('nbDims', ctypes.c_int32), ('d', ctypes.c_int32 * MAX_DIMS)]

    @staticmethod
    def of(shape: Sequence[int]) -> 'Dims':
        d = Dims()
        d.nbDims = len(shape)
        for i, s in enumerate(shape):
            d.d[i] = int(s)
        return d

    def tolist(self) -> List[int]:
        return [self.d[i] for i in range(self.nbDims)]


class TensorDesc(ctypes.Structure):
    _fields_ = [('dims', Dims), ('type', ctypes.c_int32), ('format', ctypes.c_int32), ('scale', ctypes.c_float)]


class PluginFieldC(ctypes.Structure):
    _fields_ = [('name', ctypes.c_char_p), ('data', ctypes.c_void_p), ('type', ctypes.c_int32),
                ('length', ctypes.c_int32)]


_NP_TO_PF = {
    np.dtype(np.float16): PF_FLOAT16,
    np.dtype(np.float32): PF_FLOAT32,
    np.dtype(np.float64): PF_FLOAT64,
    np.dtype(np.int8): PF_INT8,
    np.dtype(np.int16): PF_INT16,
    np.dtype(np.int32): PF_INT32,
}


class PluginField:
    """trt.PluginField(name, np.ndarray, trt.PluginFieldType) look-alike."""

    def __init__(self, name: str, data: np.ndarray, type: Optional[int] = None):
        self.name = name
        self.data = np.ascontiguousarray(data)
        self.type = _NP_TO_PF[self.data.dtype] if type is None else type


_lib = None


def lib_path() -> str:
    return os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(__file__))), 'libs', _LIB_NAME)


def load_library():
    """ctypes.CDLL(<pkg>/libs/libnvinfer_plugin_tensorrt_llm.so, RTLD_GLOBAL) + initLibNvInferPlugins
    (T/tensorrt_llm/plugin/plugin.py:7-22)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f'{path} not found: build it with `make -C trtllm-llama_amd/csrc` '
                           f'(or __graft_entry__.build()); there is no CPU fallback')
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    c = ctypes
    lib.initLibNvInferPlugins.argtypes = [c.c_void_p, c.c_char_p]
    lib.initLibNvInferPlugins.restype = c.c_bool  # the reference's signature (P/api/InferPlugin.cpp:149)
    lib.getInferLibVersion.restype = c.c_int32
    lib.tllm_last_error.restype = c.c_char_p
    lib.tllm_plugin_registry_size.restype = c.c_int32
    lib.tllm_plugin_registry_name.argtypes = [c.c_int32]
    lib.tllm_plugin_registry_name.restype = c.c_char_p
    lib.tllm_plugin_create.argtypes = [c.c_char_p, c.c_char_p, c.c_char_p, c.POINTER(PluginFieldC), c.c_int32]
    lib.tllm_plugin_create.restype = c.c_void_p
    lib.tllm_plugin_type.argtypes = [c.c_void_p]
    lib.tllm_plugin_type.restype = c.c_char_p
    lib.tllm_plugin_version.argtypes = [c.c_void_p]
    lib.tllm_plugin_version.restype = c.c_char_p
    lib.tllm_plugin_nb_outputs.argtypes = [c.c_void_p]
    lib.tllm_plugin_nb_outputs.restype = c.c_int32
    lib.tllm_plugin_output_dims.argtypes = [c.c_void_p, c.c_int32, c.POINTER(Dims), c.c_int32, c.POINTER(Dims)]
    lib.tllm_plugin_output_dims.restype = c.c_int32
    lib.tllm_plugin_output_dtype.argtypes = [c.c_void_p, c.c_int32, c.POINTER(c.c_int32), c.c_int32]
    lib.tllm_plugin_output_dtype.restype = c.c_int32
    lib.tllm_plugin_supports_format.argtypes = [c.c_void_p, c.c_int32, c.POINTER(TensorDesc), c.c_int32, c.c_int32]
    lib.tllm_plugin_supports_format.restype = c.c_int32
    lib.tllm_plugin_workspace_size.argtypes = [c.c_void_p, c.POINTER(TensorDesc), c.c_int32, c.POINTER(TensorDesc),
                                               c.c_int32]
    lib.tllm_plugin_workspace_size.restype = c.c_size_t
    lib.tllm_plugin_enqueue.argtypes = [c.c_void_p, c.POINTER(TensorDesc), c.POINTER(TensorDesc),
                                        c.POINTER(c.c_void_p), c.POINTER(c.c_void_p), c.c_void_p, c.c_void_p]
    lib.tllm_plugin_enqueue.restype = c.c_int32
    lib.tllm_plugin_serialization_size.argtypes = [c.c_void_p]
    lib.tllm_plugin_serialization_size.restype = c.c_size_t
    lib.tllm_plugin_serialize.argtypes = [c.c_void_p, c.c_void_p]
    lib.tllm_plugin_serialize.restype = c.c_int32
    lib.tllm_plugin_deserialize.argtypes = [c.c_char_p, c.c_void_p, c.c_size_t]
    lib.tllm_plugin_deserialize.restype = c.c_void_p
    lib.tllm_plugin_clone.argtypes = [c.c_void_p]
    lib.tllm_plugin_clone.restype = c.c_void_p
    lib.tllm_plugin_destroy.argtypes = [c.c_void_p]
    lib.tllm_plugin_destroy.restype = None
    lib.tllm_comm_get_unique_id.argtypes = [c.c_void_p]
    lib.tllm_comm_get_unique_id.restype = c.c_int32
    lib.tllm_comm_init_rank.argtypes = [c.POINTER(c.c_int32), c.c_int32, c.c_int32, c.c_void_p]
    lib.tllm_comm_init_rank.restype = c.c_int32
    lib.tllm_comm_destroy_all.restype = c.c_int32
    lib.tllm_symmetric_quantize_last_axis.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_int32, c.c_void_p,
                                                      c.c_void_p, c.c_void_p]
    lib.tllm_symmetric_quantize_last_axis.restype = c.c_int32
    lib.tllm_preprocess_weights_for_mixed_gemm.argtypes = [c.c_void_p, c.c_int64, c.c_int64, c.c_int32, c.c_void_p]
    lib.tllm_preprocess_weights_for_mixed_gemm.restype = c.c_int32
    if not lib.initLibNvInferPlugins(None, TRT_LLM_PLUGIN_NAMESPACE.encode('utf-8')):
        raise RuntimeError('initLibNvInferPlugins failed')
    _lib = lib
    return lib


def last_error() -> str:
    return load_library().tllm_last_error().decode('utf-8', 'replace')


def registered_plugins() -> List[str]:
    lib = load_library()
    return [lib.tllm_plugin_registry_name(i).decode() for i in range(lib.tllm_plugin_registry_size())]


class Plugin:
    """One plugin instance (IPluginV2DynamicExt look-alike over the flat C ABI)."""

    def __init__(self, handle: int, name: str):
        self._h = handle
        self.name = name

    @staticmethod
    def create(name: str, fields: Sequence[PluginField], version: str = '1',
               namespace: str = TRT_LLM_PLUGIN_NAMESPACE) -> Optional['Plugin']:
        lib = load_library()
        arr = (PluginFieldC * max(len(fields), 1))()
        keep = []
        for i, f in enumerate(fields):
            nm = f.name.encode()
            keep.append(nm)
            arr[i].name = nm
            arr[i].data = f.data.ctypes.data
            arr[i].type = f.type
            arr[i].length = int(f.data.size)
        h = lib.tllm_plugin_create(name.encode(), version.encode(), namespace.encode(), arr, len(fields))
        return Plugin(h, name) if h else None

    @staticmethod
    def deserialize(name: str, blob: bytes) -> Optional['Plugin']:
        lib = load_library()
        buf = ctypes.create_string_buffer(blob, len(blob))
        h = lib.tllm_plugin_deserialize(name.encode(), buf, len(blob))
        return Plugin(h, name) if h else None

    def serialize(self) -> bytes:
        lib = load_library()
        n = lib.tllm_plugin_serialization_size(self._h)
        buf = ctypes.create_string_buffer(max(n, 1))
        if lib.tllm_plugin_serialize(self._h, buf):
            raise RuntimeError(last_error())
        return buf.raw[:n]

    def clone(self) -> 'Plugin':
        return Plugin(load_library().tllm_plugin_clone(self._h), self.name)

    @property
    def plugin_type(self) -> str:
        return load_library().tllm_plugin_type(self._h).decode()

    @property
    def num_outputs(self) -> int:
        return load_library().tllm_plugin_nb_outputs(self._h)

    def output_dims(self, index: int, input_shapes: Sequence[Sequence[int]]) -> List[int]:
        lib = load_library()
        ins = (Dims * len(input_shapes))(*[Dims.of(s) for s in input_shapes])
        out = Dims()
        if lib.tllm_plugin_output_dims(self._h, index, ins, len(input_shapes), ctypes.byref(out)):
            raise RuntimeError(last_error())
        return out.tolist()

    def output_dtype(self, index: int, input_types: Sequence[int]) -> int:
        arr = (ctypes.c_int32 * len(input_types))(*input_types)
        return load_library().tllm_plugin_output_dtype(self._h, index, arr, len(input_types))

    @staticmethod
    def _descs(shapes, types):
        arr = (TensorDesc * max(len(shapes), 1))()
        for i, (s, t) in enumerate(zip(shapes, types)):
            arr[i].dims = Dims.of(s)
            arr[i].type = t
            arr[i].format = 0
            arr[i].scale = 1.0
        return arr

    def supports_format(self, pos, shapes, types, nb_inputs) -> bool:
        arr = self._descs(shapes, types)
        return bool(load_library().tllm_plugin_supports_format(self._h, pos, arr, nb_inputs, len(shapes) - nb_inputs))

    def workspace_size(self, in_shapes, in_types, out_shapes, out_types) -> int:
        return load_library().tllm_plugin_workspace_size(self._h, self._descs(in_shapes, in_types), len(in_shapes),
                                                         self._descs(out_shapes, out_types), len(out_shapes))

    def enqueue(self, in_shapes, in_types, in_ptrs, out_shapes, out_types, out_ptrs, workspace_ptr: int,
                stream: int) -> None:
        """Raw-pointer enqueue; raises RuntimeError('Executing TRT engine failed!') semantics on failure
        (T/tensorrt_llm/runtime/generation.py:892-894)."""
        lib = load_library()
        ind = self._descs(in_shapes, in_types)
        outd = self._descs(out_shapes, out_types)
        ins = (ctypes.c_void_p * max(len(in_ptrs), 1))(*[ctypes.c_void_p(p) for p in in_ptrs])
        outs = (ctypes.c_void_p * max(len(out_ptrs), 1))(*[ctypes.c_void_p(p) for p in out_ptrs])
        rc = lib.tllm_plugin_enqueue(self._h, ind, outd, ins, outs, ctypes.c_void_p(workspace_ptr),
                                     ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f'{self.name} enqueue failed: {last_error()}')

    def destroy(self):
        if self._h:
            load_library().tllm_plugin_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def symmetric_quantize_last_axis(weight_kn_fp16: np.ndarray, bits: int):
    """torch.ops.fastertransformer.symmetric_quantize_last_axis_of_batched_matrix equivalent
    (T/cpp/tensorrt_llm/thop/weightOnlyQuantOp.cpp:143-236).  Returns (processed int8, scales fp16, unprocessed)."""
    lib = load_library()
    w = np.ascontiguousarray(weight_kn_fp16, dtype=np.float16)
    k, n = w.shape
    row_bytes = ((k + 15) // 16 * 16) if bits == 8 else ((k + 31) // 32 * 32) // 2
    processed = np.empty((n, row_bytes), dtype=np.int8)
    scales = np.empty((n, ), dtype=np.float16)
    unprocessed = np.empty((k, n if bits == 8 else n // 2), dtype=np.int8)
    rc = lib.tllm_symmetric_quantize_last_axis(w.ctypes.data, k, n, bits, processed.ctypes.data, scales.ctypes.data,
                                               unprocessed.ctypes.data)
    if rc:
        raise RuntimeError(last_error())
    return processed, scales, unprocessed


def preprocess_weights_for_mixed_gemm(quantized_kn: np.ndarray, k: int, n: int, bits: int) -> np.ndarray:
    lib = load_library()
    q = np.ascontiguousarray(quantized_kn, dtype=np.int8)
    row_bytes = ((k + 15) // 16 * 16) if bits == 8 else ((k + 31) // 32 * 32) // 2
    processed = np.empty((n, row_bytes), dtype=np.int8)
    if lib.tllm_preprocess_weights_for_mixed_gemm(q.ctypes.data, k, n, bits, processed.ctypes.data):
        raise RuntimeError(last_error())
    return processed
