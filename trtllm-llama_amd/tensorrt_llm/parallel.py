"""Tensor-parallel bootstrap: one process per GPU; the RCCL unique id travels through torch.distributed (replaces the
MPI_Send/Recv bootstrap of P/ncclPlugin/allreducePlugin.cpp:124-162)."""
import ctypes
import os

from .plugin import capi

_done = set()


def ensure_tp_communicator(mapping) -> None:
    key = (tuple(mapping.tp_group), mapping.rank)
    if key in _done or mapping.tp_size <= 1:
        return
    import torch
    import torch.distributed as dist
    lib = capi.load_library()
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29512')
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        dist.init_process_group(backend, rank=mapping.rank, world_size=mapping.world_size)
    dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    idbuf = torch.zeros(128, dtype=torch.uint8, device=dev)
    if mapping.rank == mapping.tp_group[0]:
        raw = (ctypes.c_char * 128)()
        if lib.tllm_comm_get_unique_id(raw):
            raise RuntimeError(capi.last_error())
        idbuf.copy_(torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8))
    dist.broadcast(idbuf, mapping.tp_group[0])
    raw = (ctypes.c_char * 128).from_buffer_copy(bytes(idbuf.cpu().numpy().tobytes()))
    group = (ctypes.c_int32 * mapping.tp_size)(*mapping.tp_group)
    if lib.tllm_comm_init_rank(group, mapping.tp_size, mapping.rank, raw):
        raise RuntimeError(capi.last_error())
    _done.add(key)
