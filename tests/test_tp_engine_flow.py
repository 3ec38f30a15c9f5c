"""BASELINE.json's tensor-parallel configuration through the user-facing flow, on one GPU: hf_llama_convert.py -tp 2 (per-rank
SmoothQuant int8 weights + scales in the reference's file format) -> build.py --world_size 2 (one engine per rank) ->
two rank processes, each a GenerationSession on ITS engine, collectives on the one-shot peer-to-peer transport.  Must
agree with the tp = 1 engine of the same checkpoint: logits within the int8 bound, same first tokens."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant')
sys.path.insert(0, os.path.join(ROOT, 'tests'))
pytestmark = pytest.mark.gpu

FLAGS = ['--use_smooth_quant', '--per_channel', '--int8_kv_cache']
B, S, NEW = 2, 24, 6


def prompts():
    import torch
    g = torch.Generator().manual_seed(5)
    return torch.randint(3, 160, (B, S), generator=g).numpy().astype(np.int32), np.array([S, S - 7], np.int32)


def build(tmp_path, hf_dir, tp):
    out = tmp_path / f'ft{tp}'
    subprocess.run([sys.executable, os.path.join(EX, 'hf_llama_convert.py'), '-i', hf_dir, '-o', str(out), '-tp', str(tp), '-sq',
                    '0.5', '--calibrate-kv-cache', '--calib-samples', '8', '--calib-len', '64'], check=True, cwd=EX, timeout=600)
    eng = tmp_path / f'eng{tp}'
    subprocess.run([sys.executable, os.path.join(EX, 'build.py'), '--model_dir', str(out / f'{tp}-gpu'), '--output_dir', str(eng),
                    '--world_size', str(tp), '--max_batch_size', str(B), '--max_input_len', str(S), '--max_output_len', str(NEW),
                    '--log_level', 'error'] + FLAGS, check=True, cwd=EX, timeout=600)
    return eng


def run_engine(eng, tp, rank):
    from tensorrt_llm import Mapping
    from tensorrt_llm.runtime import GenerationSession, ModelConfig
    blob = open(eng / f'llama_float16_tp{tp}_rank{rank}.engine', 'rb').read()
    sess = GenerationSession(ModelConfig(vocab_size=160, num_layers=2, num_heads=4 // tp, hidden_size=128 // tp), blob,
                             Mapping(tp, rank))
    ids, lens = prompts()
    sess.setup(B, S, NEW)
    sess.runtime.context(ids, lens)
    l0 = sess.runtime.logits()
    sess.runtime.step(1, use_graph=False)
    l1 = sess.runtime.logits()
    sess.runtime.step(NEW - 2, use_graph=True)
    out = sess.runtime.output_ids()
    return l0, l1, out


def _rank(rank, world, port, eng, q):
    import ctypes
    import torch
    import torch.distributed as dist
    os.environ['TLLM_TEST_SHARED_GPU'] = '1'  # ranks share the GPU: RCCL refuses that, the peer-to-peer transport carries it
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
        lib.tllm_comm_p2p_enable.argtypes = [ctypes.c_int32]
        lib.tllm_comm_p2p_enable.restype = None
        h = (ctypes.c_char * 64)()
        assert lib.tllm_comm_p2p_create(world, rank, 64 * 1024, h) == 0, capi.last_error()
        allh = [torch.zeros(64, dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(allh, torch.frombuffer(bytearray(h.raw), dtype=torch.uint8))
        blob = b''.join(bytes(x.numpy().tobytes()) for x in allh)
        assert lib.tllm_comm_p2p_attach(ctypes.create_string_buffer(blob, len(blob))) == 0, capi.last_error()
        lib.tllm_comm_p2p_enable(1)
        res = run_engine(eng, world, rank)
        q.put((rank, ) + res + (lib.tllm_comm_p2p_error(), ))
        dist.barrier()
        lib.tllm_comm_destroy_all()
    except BaseException as e:
        q.put((rank, repr(e)))
        raise
    finally:
        dist.destroy_process_group()


def test_tp2_smoothquant_engines_from_the_converter(tmp_path):
    import torch.multiprocessing as mp
    from test_convert import tiny_hf
    from test_tp_session_p2p import _free_port
    sys.path.insert(0, os.path.join(ROOT, 'trtllm-llama_amd'))
    m, hf_dir = tiny_hf(tmp_path)
    eng1, eng2 = build(tmp_path, hf_dir, 1), build(tmp_path, hf_dir, 2)
    assert (eng2 / 'llama_float16_tp2_rank0.engine').exists() and (eng2 / 'llama_float16_tp2_rank1.engine').exists()
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, eng2, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    assert all(len(r) == 5 for r in res), [r for r in res if len(r) != 5]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = sorted(res, key=lambda r: r[0])
    ref0, ref1, ref_out = run_engine(eng1, 1, 0)
    scale = max(np.abs(ref0).max(), 1.0)
    for rank, l0, l1, out, err in res:
        assert err == 0
        # per-rank activation scales are the same calibration statistics and the int8 products are exact; what differs is the
        # fp16 rounding of the partial O / down outputs before the all-reduce -> an occasional 1-LSB flip in a later quantiser
        np.testing.assert_allclose(l0, ref0, atol=6e-2 * scale)
        np.testing.assert_allclose(l1, ref1, atol=6e-2 * scale)
        assert np.abs(l0 - ref0).mean() < 1e-2 * scale
        np.testing.assert_array_equal(out[:, :S], ref_out[:, :S])
    np.testing.assert_array_equal(res[0][3], res[1][3])
    # and against HF fp32 on the CPU, the bound test_convert.py uses for the single-GPU SmoothQuant engine
    import torch
    ids, lens = prompts()
    with torch.no_grad():
        hf = m(torch.from_numpy(ids[:1].astype(np.int64))).logits[:, -1].numpy()
    assert np.abs(res[0][1][:1] - hf).max() < 6e-2 * np.abs(hf).max()
