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
    if os.environ.get('TLLM_TEST_SHARED_GPU') == '1':
        # test rig: several ranks on ONE GPU (RCCL refuses that) - the peer-to-peer transport carries every collective
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


def enable_p2p_allreduce(mapping, max_bytes: int = 64 * 1024, iters: int = 8, verbose: bool = True) -> bool:
    """Switch the decode step's small fp16 all-reduces to the one-shot peer-to-peer kernel (include/tllm_plugin_api.h,
    tllm_comm_p2p_*) - but only after it has reproduced RCCL's result on THIS machine: every rank runs `iters` random
    all-reduces both ways, the verdicts are AND-ed over the ranks, and any mismatch, time-out or missing capability
    (hipIpc, peer access) leaves RCCL in charge.  TLLM_ALLREDUCE=rccl skips the attempt."""
    if mapping.tp_size <= 1 or os.environ.get('TLLM_ALLREDUCE', '').lower() == 'rccl':
        return False
    import torch
    import torch.distributed as dist
    ensure_tp_communicator(mapping)
    lib = capi.load_library()
    lib.tllm_comm_p2p_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
    lib.tllm_comm_p2p_attach.argtypes = [ctypes.c_void_p]
    lib.tllm_comm_p2p_all_reduce.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    lib.tllm_comm_p2p_enable.argtypes = [ctypes.c_int32]
    lib.tllm_comm_p2p_enable.restype = ctypes.c_int32
    dev = torch.device('cuda', torch.cuda.current_device())
    world, rank = mapping.tp_size, mapping.tp_group.index(mapping.rank)
    ok = True
    why = ''
    h = (ctypes.c_char * 64)()
    if lib.tllm_comm_p2p_create(world, rank, max_bytes, h):
        ok, why = False, capi.last_error()
    mine = torch.frombuffer(bytearray(h.raw), dtype=torch.uint8).to(dev)
    allh = [torch.zeros(64, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(allh, mine)  # every rank takes part even if its own create failed
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        blob = b''.join(bytes(t.cpu().numpy().tobytes()) for t in allh)
        if lib.tllm_comm_p2p_attach(ctypes.create_string_buffer(blob, len(blob))):
            ok, why = False, capi.last_error()
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        stream = torch.cuda.current_stream().cuda_stream
        g = torch.Generator(device='cpu').manual_seed(1234 + rank)
        for it in range(iters):  # no early exit: every rank issues the same collectives whatever it observes
            n = (4096, 8192, 32768, 520 * 8)[it % 4]
            x = (torch.randn(n, generator=g) * 3).to(torch.float16).to(dev)
            ref = x.clone()
            dist.all_reduce(ref)  # RCCL through torch
            got = x.clone()
            if lib.tllm_comm_p2p_all_reduce(got.data_ptr(), n, stream):
                ok, why = False, capi.last_error()
            torch.cuda.synchronize()
            # RCCL adds in fp16 along its ring, the one-shot kernel in fp32 in rank order: a few fp16 ulps apart at most
            if ok and not torch.allclose(got.float(), ref.float(), rtol=4e-3, atol=4e-3 * world):
                ok, why = False, f'mismatch vs RCCL at iteration {it}: max |diff| {(got.float() - ref.float()).abs().max().item():.4g}'
        if ok and lib.tllm_comm_p2p_error() != 0:
            ok, why = False, 'a flag wait timed out'
        # the fused layer seam (all-reduce + residual add + next RMSNorm + SmoothQuant quantiser in one launch,
        # kernels/p2p_allreduce.hip) is validated the same way before the decode step may use it: x <- x + sum of the partials
        # against RCCL's sum, the normalised row against torch.  A mismatch leaves the all-reduce peer-to-peer but keeps the
        # three-stage seam (tllm_comm_p2p_enable_fused(0); TLLM_NO_FUSED_ALLREDUCE=1 is the user's A/B switch, read per step).
        # (the verdict so far is made collective first: every rank must take the same path through the collectives below)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_ok = int(flag.item()) == 1
        fused_ok, fused_why = all_ok, ''
        if all_ok:
            lib.tllm_comm_p2p_all_reduce_residual_norm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
            lib.tllm_comm_p2p_state.restype = ctypes.c_int32
            gshared = torch.Generator(device='cpu').manual_seed(99)  # residual stream and gamma: identical on every rank
            # every tail the decode step may use: fp16 norm (quant 0), static int8 (1), per-token int8 + scales (2)
            cases = [(rows, cols, quant) for rows, cols in ((1, 4096), (2, 1024)) for quant in (0, 1, 2)]
            # the argument verdict is made collective BEFORE anything is launched: a rank whose call would be refused on the
            # host (not attached, row too long for a slot) must not leave its peers spinning to the time-out
            pre = all((lib.tllm_comm_p2p_state() & 1) and rows * cols * 2 <= max_bytes and cols % 8 == 0 for rows, cols, _ in cases)
            flag = torch.tensor([1 if pre else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                fused_ok, fused_why, cases = False, 'the fused seam\'s arguments are refused on a rank', []
            for rows, cols, quant in cases:
                part = (torch.randn(rows, cols, generator=g) * 0.5).to(torch.float16).to(dev)
                x0 = (torch.randn(rows, cols, generator=gshared) * 2).to(torch.float16).to(dev)
                gamma = (1 + 0.1 * torch.randn(cols, generator=gshared)).to(torch.float16).to(dev)
                qscale = torch.tensor([37.5], dtype=torch.float32, device=dev)
                total = part.clone()
                dist.all_reduce(total)
                x = x0.clone()
                out = torch.empty(rows, cols, dtype=torch.float16 if quant == 0 else torch.int8, device=dev)
                dyn = torch.zeros(rows, dtype=torch.float32, device=dev)
                if lib.tllm_comm_p2p_all_reduce_residual_norm(part.data_ptr(), x.data_ptr(), gamma.data_ptr(), 1e-6, rows, cols,
                                                              out.data_ptr(), quant, qscale.data_ptr() if quant == 1 else None,
                                                              dyn.data_ptr() if quant == 2 else None, stream):
                    fused_ok, fused_why = False, capi.last_error()
                torch.cuda.synchronize()
                want_x = (x0.float() + total.float())
                xf = x.float()
                # the kernel's rounding points: fp16(x * rsqrt) then fp16(. * gamma)   (PY/functional.py:3195-3219)
                want_n = ((xf * torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-6)).half().float() * gamma.float()).half().float()
                good = torch.allclose(xf, want_x, rtol=4e-3, atol=4e-3 * world)
                if quant == 0:
                    good = good and torch.allclose(out.float(), want_n, rtol=4e-3, atol=4e-3)
                else:
                    if quant == 1:
                        sc = qscale.expand(rows)[:, None]
                    else:  # amax / 127 per token (K/quantization.cu:94-118)
                        amax = want_n.abs().amax(-1).clamp_min(1e-6)
                        good = good and torch.allclose(dyn, amax / 127.0, rtol=2e-3)
                        sc = (127.0 / amax)[:, None]
                    want_q = torch.clamp(torch.round(want_n * sc), -128, 127)
                    good = good and int((out.float() - want_q).abs().max().item()) <= 1  # one fp16 ulp ahead of the rounding
                if fused_ok and not good:
                    fused_ok, fused_why = False, f'the fused residual + RMSNorm tail (quant {quant}) does not reproduce torch / RCCL'
            if lib.tllm_comm_p2p_error() != 0:
                ok, why = False, 'a flag wait timed out'
        flag = torch.tensor([1 if (ok and all_ok) else 0, 1 if fused_ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        fused_use = int(flag[1].item()) == 1
        if int(flag[0].item()) == 1 and not fused_use:
            why = f'fused seam off: {fused_why}' if fused_why else 'fused seam off (a peer\'s validation failed)'
        # the verdict lives in the transport's state (sessions read it per step), not in an environment variable
        lib.tllm_comm_p2p_enable_fused.argtypes = [ctypes.c_int32]
        lib.tllm_comm_p2p_enable_fused.restype = None
        lib.tllm_comm_p2p_enable_fused(1 if fused_use else 0)
        flag = flag[:1]
    use = int(flag.item()) == 1
    if lib.tllm_comm_p2p_enable(1 if use else 0) and use:
        raise RuntimeError(capi.last_error())
    if verbose and mapping.rank == mapping.tp_group[0]:
        import sys
        print(f'[tensorrt_llm.parallel] decode all-reduce: {"one-shot peer-to-peer (validated against RCCL)" if use else "RCCL"}'
              + (f' ({why})' if why else ''), file=sys.stderr)
    return use
