"""The tensor-parallel layer seam as ONE launch (all-reduce + residual add + next RMSNorm + SmoothQuant quantiser,
kernels/p2p_allreduce.hip) against the three-stage seam it replaces (TLLM_NO_FUSED_ALLREDUCE=1: rank 0 carries the residual, plain
all-reduce, the consuming GEMV normalises) and against the un-sharded session, for every operand type the consuming GEMVs take
behind it: fp16 (PRO_NONE on fp16), weight-only int8 (the splice-bias sums of a pre-normalised row), SmoothQuant static (int8 row) and
SmoothQuant per token (int8 row + one scale per row handed to the GEMV) - and with more than 8 sequences (slabs of 8 rows behind a
seam of 10 rows).  Two rank processes share the GPU (gloo for the bootstrap, the peer-to-peer transport for the model)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant')
pytestmark = pytest.mark.gpu
NEW = 6


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def shard(et, tp, rank, num_layers):
    """un-sharded engine tensors of any quantisation mode -> this rank's (T/examples/llama/weight.py:86-172 rules; per-channel
    factors follow the columns of column-parallel GEMMs, Q/convert.py:125-141)"""
    sys.path.insert(0, EX)
    import weight as W
    out = {k: v for k, v in et.items()}
    out['lm_head.weight'] = W.split(et['lm_head.weight'], tp, rank)
    for i in range(num_layers):
        p = f'layers.{i}.'
        out[p + 'attention.qkv.weight'] = W.split_qkv(et[p + 'attention.qkv.weight'], tp, rank)
        out[p + 'attention.dense.weight'] = W.split(et[p + 'attention.dense.weight'], tp, rank, dim=1)
        out[p + 'mlp.fc.weight'] = W.split(et[p + 'mlp.fc.weight'], tp, rank, dim=0)
        out[p + 'mlp.gate.weight'] = W.split(et[p + 'mlp.gate.weight'], tp, rank, dim=0)
        out[p + 'mlp.proj.weight'] = W.split(et[p + 'mlp.proj.weight'], tp, rank, dim=1)
        for n, qkv in (('attention.qkv', True), ('mlp.fc', False), ('mlp.gate', False)):
            key = p + n + '.per_channel_scale'
            if key in et and et[key].size > 1:
                flat = et[key].reshape(-1, 1)
                part = W.split_qkv(flat, tp, rank) if qkv else W.split(flat, tp, rank, dim=0)
                out[key] = part.reshape(et[key].shape[:-1] + (-1, )) if et[key].ndim == 2 else part.reshape(-1)
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def build(mode, B):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle import quant_oracle as QO
    from test_gpu_session import synth_model
    cfg, w = synth_model(57)
    r = np.random.default_rng(21)
    S = 16
    lens = r.integers(4, S + 1, B).astype(np.int32)
    lens[0] = S
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    qmodel = QO.quantise_model(cfg, w, mode, 0 if mode == 'fp16' else 1, calib_ids=ids, calib_lens=lens)
    return cfg, qmodel['engine_tensors'], qmodel['quant_mode'], ids, lens


def run(cfg, et, qm, tp, rank, ids, lens, feed=None):
    from tensorrt_llm.runtime.native import NativeSession
    B, S = ids.shape
    s = NativeSession(dict(cfg, quant_mode=qm, tp_size=tp, tp_rank=rank))
    for k, v in (shard(et, tp, rank, cfg['num_layers']) if tp > 1 else et).items():
        s.set_tensor(k, v)
    s.finalize()
    s.setup(B, S, NEW)
    s.context(ids, lens)
    logits = [s.logits()]
    for i in range(NEW - 1):
        if feed is not None:
            s.force_tokens(feed[:, i])
        s.step(1, use_graph=i > 0)
        logits.append(s.logits())
    out = s.output_ids()
    s.close()
    return np.stack(logits), out


def _rank(rank, world, port, mode, B, fused, q):
    import ctypes
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    sys.path.insert(0, ROOT)
    if not fused:
        os.environ['TLLM_NO_FUSED_ALLREDUCE'] = '1'
    from tensorrt_llm.plugin import capi
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
        lib.tllm_comm_p2p_enable(1)
        cfg, et, qm, ids, lens = build(mode, B)
        logits, out = run(cfg, et, qm, world, rank, ids, lens)
        q.put((rank, logits, out, int(lib.tllm_comm_p2p_error())))
        dist.barrier()
        lib.tllm_comm_destroy_all()
    except BaseException as e:
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('mode,B', [('fp16', 2), ('woq8', 2), ('sq_static_pc', 2), ('sq_dyn_pc', 2), ('sq_dyn_pc', 10), ('fp16', 10)])
def test_tp2_layer_seam_fused_and_three_stage(mode, B, fused):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, mode, B, fused, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    assert all(len(r) == 4 for r in res), [r for r in res if len(r) != 4]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = sorted(res, key=lambda r: r[0])
    assert res[0][3] == 0 and res[1][3] == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])  # every rank ends with bit-identical logits ...
    np.testing.assert_array_equal(res[0][2], res[1][2])  # ... and tokens
    logits, out = res[0][1], res[0][2]
    cfg, et, qm, ids, lens = build(mode, B)
    S = ids.shape[1]
    ref, ref_out = run(cfg, et, qm, 1, 0, ids, lens, feed=out[:, S:S + NEW - 1])  # un-sharded, on the sharded run's tokens
    scale = max(np.abs(ref[0]).max(), 1.0)
    sq = mode.startswith('sq')
    for i in range(NEW):
        d = np.abs(logits[i] - ref[i])
        print(f'[{mode} B={B} fused={fused}] step {i}: max |d| {d.max():.4g} mean |d| {d.mean():.4g} (scale {scale:.3g})')
        # a row-parallel GEMM split over two ranks rounds two fp16 partial products instead of one; SmoothQuant amplifies the
        # fp16 ulps through its quantisers (same bound as tests/test_tp_7b_extents.py)
        assert d.max() < (8e-2 if sq else 2e-2) * scale and d.mean() < (1.5e-2 if sq else 4e-3) * scale, (i, d.max(), d.mean())
    np.testing.assert_array_equal(out[:, :S + 1], ref_out[:, :S + 1])
