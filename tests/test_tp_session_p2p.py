"""Tensor-parallel decode END TO END on one GPU: N processes, each a rank with ITS shard of the weights in its own C++
session (heads / FFN columns split, vocabulary split), exchanging partial sums through the one-shot peer-to-peer
all-reduce / all-gather inside the captured step graph.  Must reproduce the un-sharded session: same greedy tokens,
logits within the fp16 bound of a re-ordered sum.  (On N GPUs the only difference is the transport under the same
kernels; RCCL remains the default there until tensorrt_llm.parallel.enable_p2p_allreduce has validated the path.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant')
GOLD = os.path.join(ROOT, 'tests', 'golden')
pytestmark = pytest.mark.gpu

NEW = 40
NEW_BEAM = 10


def model():
    """Seeded synthetic 2-layer LLaMA (D = 256, 4 heads, I = 512, V = 512): every per-rank extent stays 16-byte aligned at
    tp = 2 (the golden HF model's I = 24 does not)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import test_gpu_session as G
    cfg, w = G.synth_model(31)
    r = np.random.default_rng(3)
    ids = r.integers(3, cfg['vocab_size'], (2, 12)).astype(np.int32)
    lens = np.array([12, 9], np.int32)
    ids[1, 9:] = 2
    return cfg, w, ids, lens


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard(t, tp, rank):
    """golden (un-sharded, reference module names) -> this rank's engine tensors, via the loaders' split helpers."""
    sys.path.insert(0, EX)
    import weight as W
    out = {'vocab_embedding.weight': t['vocab_embedding.weight'], 'ln_f.weight': t['ln_f.weight'],
           'lm_head.weight': np.ascontiguousarray(W.split(t['lm_head.weight'], tp, rank))}
    for i in range(2):
        p = f'layers.{i}.'
        out[p + 'input_layernorm.weight'] = t[p + 'input_layernorm.weight']
        out[p + 'post_layernorm.weight'] = t[p + 'post_layernorm.weight']
        out[p + 'attention.qkv.weight'] = W.split_qkv(t[p + 'attention.qkv.weight'], tp, rank)
        out[p + 'attention.dense.weight'] = W.split(t[p + 'attention.dense.weight'], tp, rank, dim=1)
        out[p + 'mlp.fc.weight'] = W.split(t[p + 'mlp.fc.weight'], tp, rank, dim=0)
        out[p + 'mlp.gate.weight'] = W.split(t[p + 'mlp.gate.weight'], tp, rank, dim=0)
        out[p + 'mlp.proj.weight'] = W.split(t[p + 'mlp.proj.weight'], tp, rank, dim=1)
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def _rank(rank, world, port, q):
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    from tensorrt_llm.plugin import capi
    from tensorrt_llm.runtime.native import NativeSession
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        lib = capi.load_library()
        lib.tllm_comm_p2p_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
        lib.tllm_comm_p2p_attach.argtypes = [ctypes.c_void_p]
        lib.tllm_comm_p2p_enable.argtypes = [ctypes.c_int32]
        lib.tllm_comm_p2p_enable.restype = None
        h = (ctypes.c_char * 64)()
        assert lib.tllm_comm_p2p_create(world, rank, 64 * 1024, h) == 0, capi.last_error()
        allh = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(allh, torch.frombuffer(bytearray(h.raw), dtype=torch.uint8))
        blob = b''.join(bytes(x.numpy().tobytes()) for x in allh)
        assert lib.tllm_comm_p2p_attach(ctypes.create_string_buffer(blob, len(blob))) == 0, capi.last_error()
        lib.tllm_comm_p2p_enable(1)  # no RCCL communicator exists in this test: a fall-back would fail loudly
        CFG, t, ids, lens = model()
        B, S = ids.shape
        s = NativeSession(dict(CFG, quant_mode=0, tp_size=world, tp_rank=rank))
        for k, v in shard(t, world, rank).items():
            s.set_tensor(k, v)
        s.finalize()
        s.setup(B, S, NEW)
        s.context(ids, lens)
        logits_ctx = s.logits()
        s.step(1, use_graph=False)
        logits_dec = s.logits()
        s.step(NEW - 1, use_graph=True)  # captured graph with the peer-to-peer kernels inside, replayed 39 times
        out = s.output_ids()
        s.close()
        # beam search over a paged cache, sharded: the beam step reads the vocabulary-split logits [tp, batch * beam, V / tp]
        s = NativeSession(dict(CFG, quant_mode=0, tp_size=world, tp_rank=rank, paged_kv_cache=1, tokens_per_block=8))
        for k, v in shard(t, world, rank).items():
            s.set_tensor(k, v)
        s.finalize()
        s.setup(B, S, NEW_BEAM, beam_width=2)
        s.generate(ids, lens, NEW_BEAM)
        beams, cum = s.beam_output()
        s.close()
        q.put((rank, logits_ctx, logits_dec, out, lib.tllm_comm_p2p_error(), beams, cum))
        dist.barrier()
        lib.tllm_comm_destroy_all()
    except BaseException as e:  # the parent must not wait for a result that will never come
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_tp2_sessions_on_one_gpu_match_the_unsharded_session():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    from tensorrt_llm.runtime.native import NativeSession
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    assert all(len(r) == 7 for r in res), [r for r in res if len(r) != 7]
    res = sorted(res, key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # un-sharded reference run
    CFG, t, ids, lens = model()
    B, S = ids.shape
    s = NativeSession(dict(CFG, quant_mode=0))
    for k, v in shard(t, 1, 0).items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    ref_ctx = s.logits()
    s.step(1, use_graph=False)
    ref_dec = s.logits()
    s.step(NEW - 1, use_graph=True)
    ref_out = s.output_ids()
    s.close()
    s = NativeSession(dict(CFG, quant_mode=0))
    for k, v in shard(t, 1, 0).items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW_BEAM, beam_width=2)
    s.generate(ids, lens, NEW_BEAM)
    ref_beams, ref_cum = s.beam_output()
    s.close()
    # beam search: the ranks agree with each other exactly, and with the un-sharded linear-cache run up to near-ties
    np.testing.assert_array_equal(res[0][5], res[1][5])
    np.testing.assert_array_equal(res[0][6], res[1][6])
    assert res[0][5].shape == ref_beams.shape == (B, 2, S + NEW_BEAM)
    np.testing.assert_allclose(res[0][6], ref_cum, atol=0.1)
    np.testing.assert_array_equal(res[0][5][:, :, :S + 2], ref_beams[:, :, :S + 2])
    for rank, lc, ld, out, err, _, _ in res:
        assert err == 0, f'rank {rank}: a peer-to-peer wait timed out'
        scale = max(np.abs(ref_ctx).max(), 1.0)
        np.testing.assert_allclose(lc, ref_ctx, atol=2e-2 * scale)
        np.testing.assert_allclose(ld, ref_dec, atol=2e-2 * scale)
        np.testing.assert_array_equal(out[:, :S + 1], ref_out[:, :S + 1])
    # every rank ends with the same tokens (the sampler runs on identical gathered logits)
    np.testing.assert_array_equal(res[0][3], res[1][3])
    # and they follow the un-sharded greedy path except where a re-ordered fp16 sum flips a near-tie
    agree = np.mean(res[0][3][:, S:] == ref_out[:, S:])
    assert agree > 0.9, agree


def _p2p_up(lib, capi, torch, dist, ctypes, world, rank):
    h = (ctypes.c_char * 64)()
    assert lib.tllm_comm_p2p_create(world, rank, 64 * 1024, h) == 0, capi.last_error()
    allh = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(allh, torch.frombuffer(bytearray(h.raw), dtype=torch.uint8))
    blob = b''.join(bytes(x.numpy().tobytes()) for x in allh)
    assert lib.tllm_comm_p2p_attach(ctypes.create_string_buffer(blob, len(blob))) == 0, capi.last_error()
    lib.tllm_comm_p2p_enable(1)


def _timeout_rank(rank, world, port, q):
    import ctypes
    import time
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    from tensorrt_llm.plugin import capi
    from tensorrt_llm.runtime.native import NativeSession
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        lib = capi.load_library()
        lib.tllm_comm_p2p_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p]
        lib.tllm_comm_p2p_attach.argtypes = [ctypes.c_void_p]
        lib.tllm_comm_p2p_enable.argtypes = [ctypes.c_int32]
        lib.tllm_comm_p2p_enable.restype = None
        lib.tllm_comm_p2p_set_max_spins.argtypes = [ctypes.c_int32]
        lib.tllm_comm_p2p_set_max_spins.restype = None
        _p2p_up(lib, capi, torch, dist, ctypes, world, rank)
        lib.tllm_comm_p2p_set_max_spins(20000)  # a few milliseconds
        CFG, t, ids, lens = model()
        B, S = ids.shape

        def session():
            s = NativeSession(dict(CFG, quant_mode=0, tp_size=world, tp_rank=rank))
            for k, v in shard(t, world, rank).items():
                s.set_tensor(k, v)
            s.finalize()
            s.setup(B, S, 8)
            return s

        s = session()
        msgs = []
        if rank == 1:
            time.sleep(1.5)  # rank 0's first all-reduce gives up long before this rank shows up
        for attempt in range(2):
            try:
                s.context(ids, lens)
                s.step(2)
                s.logits()
                msgs.append('ok')
            except RuntimeError as e:
                msgs.append(str(e))
        s.close()
        dist.barrier()
        # a fresh transport (destroy + create + attach + enable) serves a fresh session
        lib.tllm_comm_destroy_all()
        _p2p_up(lib, capi, torch, dist, ctypes, world, rank)
        s = session()
        out = s.generate(ids, lens, 8, end_id=-1)
        s.close()
        q.put((rank, msgs, out, int(lib.tllm_comm_p2p_error())))
        dist.barrier()
        lib.tllm_comm_destroy_all()
    except BaseException as e:
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_a_session_call_after_a_timeout_is_not_failed_by_the_sticky_flag():
    """ADVICE r2 (medium), session level.  Rank 1 is late; rank 0's first peer-to-peer all-reduce gives up.  The call fails on
    BOTH ranks with the time-out error (rank 1 learns of it through the poison word), both take the transport out of service,
    and the NEXT call is no longer failed by the recorded time-out: it goes to the fall-back transport (RCCL; none exists on this
    one-GPU rig, so it fails with 'no communicator', not with the time-out).  A re-created transport then serves a new session."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timeout_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(len(r) == 4 for r in res), res
    res = sorted(res, key=lambda r: r[0])
    for rank, msgs, out, err in res:
        assert 'timed out' in msgs[0], (rank, msgs)
        assert ('on this rank' in msgs[0]) == (rank == 0) and ('reported by a peer' in msgs[0]) == (rank == 1), (rank, msgs)
        assert 'timed out' not in msgs[1] and 'communicator' in msgs[1], (rank, msgs)
        assert err == 0, (rank, err)
    np.testing.assert_array_equal(res[0][2], res[1][2])
