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


def _bootstrap(rank, world, port, max_bytes=64 * 1024):
    """process-group + inbox exchange shared by the workers below; returns (lib, capi, torch, dist, stream)"""
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    from tensorrt_llm.plugin import capi
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lib = capi.load_library()
    lib.tllm_comm_p2p_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
    lib.tllm_comm_p2p_attach.argtypes = [ctypes.c_void_p]
    lib.tllm_comm_p2p_all_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.tllm_comm_p2p_all_reduce_residual_norm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int32,
                                                           ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                                           ctypes.c_void_p]
    lib.tllm_comm_p2p_set_max_spins.argtypes = [ctypes.c_int32]
    lib.tllm_comm_p2p_set_max_spins.restype = None
    lib.tllm_comm_p2p_enable.argtypes = [ctypes.c_int32]
    lib.tllm_comm_p2p_enable.restype = None
    h = (ctypes.c_char * 64)()
    assert lib.tllm_comm_p2p_create(world, rank, max_bytes, h) == 0, capi.last_error()
    allh = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(allh, torch.frombuffer(bytearray(h.raw), dtype=torch.uint8))
    blob = b''.join(bytes(t.numpy().tobytes()) for t in allh)
    assert lib.tllm_comm_p2p_attach(ctypes.create_string_buffer(blob, len(blob))) == 0, capi.last_error()
    return lib, capi, torch, dist, torch.cuda.current_stream().cuda_stream


def _fused_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    try:
        lib, capi, torch, dist, stream = _bootstrap(rank, world, port)
        from oracle import llama_oracle as O
        ok, worst = True, {}
        it = 0
        for rows, cols in ((1, 4096), (2, 256), (8, 4096), (3, 520 * 8)):
            for quant in (0, 1, 2):
                for rep in range(3):
                    it += 1
                    rng = lambda r: np.random.default_rng(10000 * it + r)
                    parts = [(rng(r).standard_normal((rows, cols)) * 0.5).astype(np.float16) for r in range(world)]
                    x0 = (rng(100).standard_normal((rows, cols)) * 2).astype(np.float16)
                    x0[:, 5] *= 20  # an outlier channel
                    gamma = (1 + 0.1 * rng(101).uniform(-1, 1, cols)).astype(np.float16)
                    qs = np.array([11.5], np.float32)
                    part = torch.from_numpy(parts[rank].copy()).cuda()
                    x = torch.from_numpy(x0.copy()).cuda()
                    g = torch.from_numpy(gamma).cuda()
                    out = torch.empty((rows, cols), dtype=torch.float16 if quant == 0 else torch.int8, device='cuda')
                    qsd = torch.from_numpy(qs).cuda()
                    dyn = torch.zeros(rows, dtype=torch.float32, device='cuda')
                    assert lib.tllm_comm_p2p_all_reduce_residual_norm(
                        part.data_ptr(), x.data_ptr(), g.data_ptr(), 1e-6, rows, cols, out.data_ptr(), quant,
                        qsd.data_ptr() if quant == 1 else None, dyn.data_ptr() if quant == 2 else None, stream) == 0, capi.last_error()
                    if rep == 2:
                        torch.cuda.synchronize()
                    # x: fp32 sum in rank order from 0, the residual last, one rounding - exact
                    acc = np.zeros((rows, cols), np.float32)
                    for r in range(world):
                        acc = acc + parts[r].astype(np.float32)
                    acc = acc + x0.astype(np.float32)
                    want_x = acc.astype(np.float16)
                    got_x = x.cpu().numpy()
                    ok = ok and np.array_equal(got_x, want_x)
                    assert np.array_equal(part.cpu().numpy(), parts[rank]), 'the partial is an input'
                    # norm: the oracle's RMSNorm on the exact x (fp32 statistics in another order: a rare fp16 tie may differ)
                    y = O.rmsnorm(want_x.astype(np.float32), gamma.astype(np.float32), 1e-6)
                    got = out.cpu().numpy()
                    if quant == 0:
                        d = np.abs(got.astype(np.float32) - y)
                        ulp = np.maximum(np.abs(y), 1e-3) * 2.0 ** -10
                        worst[quant] = max(worst.get(quant, 0.0), float((d / ulp).max()))
                        ok = ok and bool((d <= 1.01 * ulp).all()) and float((d != 0).mean()) < 0.01
                    else:
                        if quant == 1:
                            want_q = O.quantize_tensor(y, qs[0])
                        else:
                            want_q, sc = O.quantize_per_token(y)
                            ok = ok and np.allclose(dyn.cpu().numpy(), sc[:, 0], rtol=1e-6)
                        d = np.abs(got.astype(np.int32) - want_q.astype(np.int32))
                        worst[quant] = max(worst.get(quant, 0.0), float(d.max()))
                        ok = ok and d.max() <= 1 and float((d != 0).mean()) < 0.01
        torch.cuda.synchronize()
        q.put((rank, bool(ok), int(lib.tllm_comm_p2p_error()), worst))
        dist.barrier()
        lib.tllm_comm_destroy_all()
    except BaseException as e:
        q.put((rank, False, repr(e), {}))
        raise
    finally:
        import torch.distributed as dist
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_fused_allreduce_residual_rmsnorm_quant_between_processes(world):
    """The tensor-parallel layer seam in one launch (tllm_comm_p2p_all_reduce_residual_norm): x <- x + sum of the ranks' partials
    exactly (fp32 in rank order, one rounding); the RMSNorm / static / per-token quantiser tail against the oracle's
    (reference rounding points) to one fp16 ulp / one LSB on under 1 % of the elements; ranks run ahead of each other between syncs."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, err, worst in res:
        assert err == 0, f'rank {rank}: {err}'
        assert ok, f'rank {rank}: wrong results ({worst})'
    print('worst deviation per quant mode (fp16 ulps / LSBs):', res[0][3])


def _collective_timeout_worker(rank, world, port, q):
    import time
    try:
        lib, capi, torch, dist, stream = _bootstrap(rank, world, port)
        lib.tllm_comm_p2p_set_max_spins(20000)  # give up after a few milliseconds instead of a second
        res = {}
        x = torch.ones(4096, dtype=torch.float16, device='cuda')
        if rank == 0:
            # rank 1 is late: rank 0 gives up, raises its error word and poisons every peer
            assert lib.tllm_comm_p2p_all_reduce(x.data_ptr(), 4096, stream) == 0, capi.last_error()
            torch.cuda.synchronize()
            res['err'] = int(lib.tllm_comm_p2p_error())
        dist.barrier()
        if rank != 0:
            # ... a rank that has not launched anything yet sees the failure on the host side already,
            res['err_before_launch'] = int(lib.tllm_comm_p2p_error())
            # and its next launch returns at once, leaves its buffer alone and raises its own error word
            t0 = time.time()
            assert lib.tllm_comm_p2p_all_reduce(x.data_ptr(), 4096, stream) == 0, capi.last_error()
            torch.cuda.synchronize()
            res['late_launch_s'] = time.time() - t0
            res['err'] = int(lib.tllm_comm_p2p_error())
            res['untouched'] = bool((x == 1).all())
        q.put((rank, res))
        dist.barrier()
        lib.tllm_comm_destroy_all()
    except BaseException as e:
        q.put((rank, {'exc': repr(e)}))
        raise
    finally:
        import torch.distributed as dist
        dist.destroy_process_group()


def test_a_timeout_takes_every_rank_out_not_only_the_one_that_waited():
    """the decision to leave the transport must be collective.  Rank 0 times out waiting for rank 1 and writes
    a poison word into every peer's region; rank 1 - which never waited for anybody - sees it on the host (error word) before it
    launches anything, and its next launch backs off at once instead of exchanging with a rank that has gone back to RCCL."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_collective_timeout_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all('exc' not in r for r in res.values()), res
    assert res[0]['err'] != 0 and not res[0]['err'] & 0x80000000, res  # its own time-out
    for r in (1, 2):
        assert res[r]['err_before_launch'] & 0x80000000, res  # "reported by a peer", before any launch of its own
        assert res[r]['err'] & 0x80000000 and res[r]['untouched'] and res[r]['late_launch_s'] < 0.5, res
