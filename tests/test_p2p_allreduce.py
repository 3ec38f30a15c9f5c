"""One-shot peer-to-peer all-reduce (include/tllm_plugin_api.h: tllm_comm_p2p_*).  A 1-GPU box cannot host two RCCL
ranks, but two PROCESSES can share the GPU and map each other's inbox with hipIpc - the same handle exchange, kernel,
flag protocol and generation alternation that N GPUs run over xGMI (what differs there is the transport, which the
caller validates against RCCL before enabling the path: tensorrt_llm/parallel.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_iter, sizes, q):
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    from tensorrt_llm.plugin import capi
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        lib = capi.load_library()
        lib.tllm_comm_p2p_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
        lib.tllm_comm_p2p_attach.argtypes = [ctypes.c_void_p]
        lib.tllm_comm_p2p_all_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        h = (ctypes.c_char * 64)()
        assert lib.tllm_comm_p2p_create(world, rank, 64 * 1024, h) == 0, capi.last_error()
        mine = torch.frombuffer(bytearray(h.raw), dtype=torch.uint8)
        allh = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(allh, mine)
        blob = b''.join(bytes(t.numpy().tobytes()) for t in allh)
        assert lib.tllm_comm_p2p_attach(ctypes.create_string_buffer(blob, len(blob))) == 0, capi.last_error()
        stream = torch.cuda.current_stream().cuda_stream
        ok = True
        for it in range(n_iter):
            n = sizes[it % len(sizes)]
            # every rank can rebuild every rank's input: seeded by (iteration, rank)
            xs = [np.random.default_rng(1000 * it + r).standard_normal(n).astype(np.float16) for r in range(world)]
            want = np.sum(np.stack([x.astype(np.float32) for x in xs]), axis=0, dtype=np.float32).astype(np.float16)
            t = torch.from_numpy(xs[rank].copy()).cuda()
            assert lib.tllm_comm_p2p_all_reduce(t.data_ptr(), n, stream) == 0, capi.last_error()
            if it % 16 == 15 or it == n_iter - 1:  # no host sync in between: ranks run ahead of each other
                torch.cuda.synchronize()
            got = t.cpu().numpy()
            # fp32 accumulation in rank order, one rounding: exact against the same computation on the host
            ok = ok and np.array_equal(got, want)
        torch.cuda.synchronize()
        err = lib.tllm_comm_p2p_error()
        q.put((rank, ok, err))
        dist.barrier()
        lib.tllm_comm_destroy_all()
    except BaseException as e:
        q.put((rank, False, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_p2p_allreduce_between_processes_on_one_gpu(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    sizes = [4096, 8, 8192, 32768, 520]  # B x D of the decode step, odd sizes, the 64 KB limit
    procs = [ctx.Process(target=_worker, args=(r, world, port, 200, sizes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err in res:
        assert err == 0, f'rank {rank}: a spin timed out (epoch {err})'
        assert ok, f'rank {rank}: wrong sums'


def _enable_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        import tensorrt_llm.parallel as P
        from tensorrt_llm import Mapping
        P.ensure_tp_communicator = lambda mapping: None  # two ranks cannot share a GPU under RCCL; gloo plays its part
        used = P.enable_p2p_allreduce(Mapping(world, rank), verbose=False)
        q.put((rank, bool(used), ''))
        dist.barrier()
        from tensorrt_llm.plugin import capi
        capi.load_library().tllm_comm_destroy_all()
    except BaseException as e:
        q.put((rank, False, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_validated_enable_path_of_the_bootstrap():
    """tensorrt_llm.parallel.enable_p2p_allreduce - handle exchange, validation against the library all-reduce of
    torch.distributed, AND-ed verdict, enable - run by two processes on one GPU (gloo standing in for RCCL)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_enable_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res


def _timeout_worker(rank, world, port, q):
    import ctypes
    import time
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    from tensorrt_llm.plugin import capi
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        lib = capi.load_library()
        lib.tllm_comm_p2p_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
        lib.tllm_comm_p2p_attach.argtypes = [ctypes.c_void_p]
        lib.tllm_comm_p2p_all_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
        h = (ctypes.c_char * 64)()
        assert lib.tllm_comm_p2p_create(world, rank, 64 * 1024, h) == 0, capi.last_error()
        mine = torch.frombuffer(bytearray(h.raw), dtype=torch.uint8)
        allh = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(allh, mine)
        blob = b''.join(bytes(t.numpy().tobytes()) for t in allh)
        assert lib.tllm_comm_p2p_attach(ctypes.create_string_buffer(blob, len(blob))) == 0, capi.last_error()
        stream = torch.cuda.current_stream().cuda_stream
        res = {}
        if rank == 0:
            # rank 1 never shows up for this all-reduce
            x = torch.arange(4096, dtype=torch.float16, device='cuda')
            keep = x.clone()
            t0 = time.time()
            assert lib.tllm_comm_p2p_all_reduce(x.data_ptr(), 4096, stream) == 0, capi.last_error()
            torch.cuda.synchronize()
            res['first_s'] = time.time() - t0
            res['err'] = int(lib.tllm_comm_p2p_error())
            res['untouched'] = bool(torch.equal(x, keep))
            t0 = time.time()
            for _ in range(64):  # the rest of a replayed step graph: every launch backs off at once
                assert lib.tllm_comm_p2p_all_reduce(x.data_ptr(), 4096, stream) == 0
            torch.cuda.synchronize()
            res['later_s'] = time.time() - t0
            res['untouched_later'] = bool(torch.equal(x, keep))
        q.put((rank, res))
        dist.barrier()
        lib.tllm_comm_destroy_all()
    except BaseException as e:
        q.put((rank, {'exc': repr(e)}))
        raise
    finally:
        dist.destroy_process_group()


def test_a_timed_out_wait_is_sticky_and_leaves_the_buffer_alone():
    """a flag wait that expires must not be followed by a sum over a stale inbox.  Rank 1 skips an
    all-reduce: rank 0's launch gives up after its bounded spin, raises the error flag, leaves x as it was and does not
    advance the epoch; every later launch returns immediately (the session turns the flag into a failed call:
    runtime/session.cpp check_comm)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timeout_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = res[0]
    assert 'exc' not in r0, r0
    assert r0['err'] != 0, 'the expired wait must raise the error flag'
    assert r0['untouched'] and r0['untouched_later'], 'a timed-out all-reduce must not write sums of a stale inbox'
    assert r0['later_s'] < max(0.5, 0.2 * r0['first_s']), f'later launches must back off at once: {r0}'
