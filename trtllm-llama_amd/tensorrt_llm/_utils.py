"""dtype maps and process helpers (T/tensorrt_llm/_utils.py).  `trt.DataType` is replaced by `DataType`,
an IntEnum with nvinfer1::DataType's values (they travel into plugin fields as `type_id`)."""
import os
from enum import IntEnum

import numpy as np


class DataType(IntEnum):
    FLOAT = 0
    HALF = 1
    INT8 = 2
    INT32 = 3
    BOOL = 4
    UINT8 = 5
    FP8 = 6

    @property
    def itemsize(self):
        return {0: 4, 1: 2, 2: 1, 3: 4, 4: 1, 5: 1, 6: 1}[int(self)]


float32, float16, int8, int32, bool_ = DataType.FLOAT, DataType.HALF, DataType.INT8, DataType.INT32, DataType.BOOL

_str_to_dt = {'float32': DataType.FLOAT, 'float16': DataType.HALF, 'int8': DataType.INT8, 'int32': DataType.INT32,
              'bool': DataType.BOOL}
_str_to_np = {'float32': np.float32, 'float16': np.float16, 'int8': np.int8, 'int32': np.int32, 'bool': np.bool_}
_dt_to_np = {DataType.FLOAT: np.float32, DataType.HALF: np.float16, DataType.INT8: np.int8, DataType.INT32: np.int32,
             DataType.BOOL: np.bool_}


def str_dtype_to_trt(dtype):
    if dtype == 'bfloat16':
        raise ValueError('bfloat16 is not built for the MI355X path (fp16 storage, fp32 accumulate)')
    return _str_to_dt[dtype]


def str_dtype_to_np(dtype):
    return _str_to_np[dtype]


def trt_dtype_to_np(dtype):
    return _dt_to_np[DataType(dtype)]


def np_dtype_to_trt(dtype):
    dtype = np.dtype(dtype)
    for k, v in _dt_to_np.items():
        if np.dtype(v) == dtype:
            return k
    raise TypeError(f'unsupported numpy dtype {dtype}')


def trt_dtype_to_str(dtype):
    return {v: k for k, v in _str_to_dt.items()}[DataType(dtype)]


def str_dtype_to_torch(dtype):
    import torch
    return {'float32': torch.float32, 'float16': torch.float16, 'int8': torch.int8, 'int32': torch.int32,
            'bool': torch.bool}[dtype]


def torch_to_numpy(x):
    return x.detach().cpu().numpy()


def pad_vocab_size(vocab_size, tp_size):
    return int((vocab_size + tp_size - 1) // tp_size * tp_size)


def mpi_rank():
    """One process per GPU.  The reference reads the rank from mpi4py (T/tensorrt_llm/_utils.py:181-190); here
    torchrun / mpirun environment variables are used (mpi4py is not required)."""
    for k in ('RANK', 'OMPI_COMM_WORLD_RANK', 'PMI_RANK'):
        if k in os.environ:
            return int(os.environ[k])
    return 0


def mpi_world_size():
    for k in ('WORLD_SIZE', 'OMPI_COMM_WORLD_SIZE', 'PMI_SIZE'):
        if k in os.environ:
            return int(os.environ[k])
    return 1
