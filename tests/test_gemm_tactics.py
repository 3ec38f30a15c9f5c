"""On-device tactic selection for the prefill GEMMs (include/tllm_runtime_api.h: tllm_gemm_profile / tllm_gemm_tactics_*;
reference: the SmoothQuant GEMM plugin's per-M-bucket profile, K/cutlass_kernels/int8_gemm/int8_gemm_template.h:372-457, kept in
its serialisation, P/smoothQuantGemmPlugin/smoothQuantGemmPlugin.cpp:253-282)."""
import ctypes

import numpy as np
import pytest

from tensorrt_llm.plugin import capi


def _lib():
    lib = capi.load_library()
    lib.tllm_gemm_profile.argtypes = [ctypes.c_int32] * 4 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.tllm_gemm_tactics_export.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    lib.tllm_gemm_tactics_export.restype = ctypes.c_int64
    lib.tllm_gemm_tactics_import.argtypes = [ctypes.c_char_p]
    lib.tllm_gemm_tactic_lookup.argtypes = [ctypes.c_int32] * 4
    lib.tllm_gemm_tactics_clear.restype = None
    return lib


def _export(lib):
    n = lib.tllm_gemm_tactics_export(None, 0)
    buf = ctypes.create_string_buffer(n)
    lib.tllm_gemm_tactics_export(buf, n)
    return buf.value.decode()


def test_tactic_table_text_round_trip_and_bucket_lookup():
    """host only: import -> lookup (exact M, nearest M of the same power-of-two bucket, other buckets and shapes fall back to the
    static rule = 0) -> export -> import."""
    lib = _lib()
    lib.tllm_gemm_tactics_clear()
    assert _export(lib) == ''
    assert lib.tllm_gemm_tactics_import(b'3:1024:12288:4096:20:45.50;3:600:12288:4096:8:30.00;0:1024:4096:4096:6:40.00;') == 0
    assert lib.tllm_gemm_tactic_lookup(3, 1024, 12288, 4096) == 20
    assert lib.tllm_gemm_tactic_lookup(3, 1000, 12288, 4096) == 20  # bucket (512, 1024]: 1024 is nearer than 600
    assert lib.tllm_gemm_tactic_lookup(3, 700, 12288, 4096) == 8
    assert lib.tllm_gemm_tactic_lookup(3, 2048, 12288, 4096) == 0   # another bucket
    assert lib.tllm_gemm_tactic_lookup(3, 1024, 4096, 4096) == 0    # another shape / type
    assert lib.tllm_gemm_tactic_lookup(0, 1024, 4096, 4096) == 6
    text = _export(lib)
    lib.tllm_gemm_tactics_clear()
    assert lib.tllm_gemm_tactic_lookup(3, 1024, 12288, 4096) == 0
    assert lib.tllm_gemm_tactics_import(text.encode()) == 0 and _export(lib) == text
    assert lib.tllm_gemm_tactics_import(b'3:oops') != 0 and 'parse' in capi.last_error()
    lib.tllm_gemm_tactics_clear()


@pytest.mark.gpu
@pytest.mark.parametrize('wtype,m,n,k', [(3, 1024, 4096, 4096), (3, 300, 456, 1152), (0, 256, 1024, 512)])
def test_profile_picks_a_kernel_and_the_gemm_stays_exact(wtype, m, n, k):
    """tllm_gemm_profile times the candidate kernels on the device and records the fastest; the GEMM launched afterwards runs that
    kernel (tllm_gemm_tactic_lookup) and its result is the oracle's - whichever tile shape / pipeline won on this box."""
    import torch
    from oracle import llama_oracle as O
    lib = _lib()
    lib.tllm_gemm_tactics_clear()
    cfg, us = ctypes.c_int32(0), ctypes.c_float(0)
    assert lib.tllm_gemm_profile(wtype, m, n, k, ctypes.byref(cfg), ctypes.byref(us), None) == 0, capi.last_error()
    assert cfg.value > 0 and 0 < us.value < 1e5
    assert lib.tllm_gemm_tactic_lookup(wtype, m, n, k) == cfg.value
    print(f'wtype {wtype} {m} x {n} x {k}: kernel id {cfg.value}, {us.value:.1f} us; table: {_export(lib)}')

    class GemmParams(ctypes.Structure):
        _fields_ = [('wtype', ctypes.c_int32), ('out_dtype', ctypes.c_int32), ('M', ctypes.c_int32), ('N', ctypes.c_int32),
                    ('K', ctypes.c_int32), ('a', ctypes.c_void_p), ('lda', ctypes.c_int64), ('w', ctypes.c_void_p),
                    ('ldw', ctypes.c_int64), ('scale_col', ctypes.c_void_p), ('scale_row', ctypes.c_void_p),
                    ('per_channel', ctypes.c_int32), ('per_token', ctypes.c_int32), ('c', ctypes.c_void_p), ('ldc', ctypes.c_int64)]

    lib.tllm_gemm.argtypes = [ctypes.POINTER(GemmParams), ctypes.c_void_p]
    r = np.random.default_rng(1)
    c = torch.empty((m, n), dtype=torch.float16, device='cuda')
    if wtype == 3:
        a = torch.from_numpy(r.integers(-128, 128, (m, k)).astype(np.int8)).cuda()
        w = torch.from_numpy(r.integers(-128, 128, (n, k)).astype(np.int8)).cuda()
        sa = torch.from_numpy((r.uniform(0.5, 1.5, m) * 1e-2).astype(np.float32)).cuda()
        sb = torch.from_numpy((r.uniform(0.5, 1.5, n) * 1e-2).astype(np.float32)).cuda()
        q = GemmParams(3, 1, m, n, k, a.data_ptr(), k, w.data_ptr(), k, sb.data_ptr(), sa.data_ptr(), 1, 1, c.data_ptr(), n)
        assert lib.tllm_gemm(ctypes.byref(q), None) == 0, capi.last_error()
        torch.cuda.synchronize()
        ref = O.sq_gemm(a.cpu().numpy(), w.cpu().numpy(), sa.cpu().numpy(), sb.cpu().numpy())
        np.testing.assert_array_equal(c.cpu().numpy().astype(np.float32), ref)
    else:
        a = torch.from_numpy(r.standard_normal((m, k)).astype(np.float16)).cuda()
        w = torch.from_numpy((r.standard_normal((n, k)) / np.sqrt(k)).astype(np.float16)).cuda()
        q = GemmParams(0, 1, m, n, k, a.data_ptr(), k, w.data_ptr(), 2 * k, None, None, 0, 0, c.data_ptr(), n)
        assert lib.tllm_gemm(ctypes.byref(q), None) == 0, capi.last_error()
        torch.cuda.synchronize()
        ref = O.gemm_fp16(a.cpu().numpy().astype(np.float32), w.cpu().numpy().astype(np.float32))
        np.testing.assert_allclose(c.cpu().numpy().astype(np.float32), ref, rtol=2e-3, atol=2e-3)
    lib.tllm_gemm_tactics_clear()


def test_a_session_config_carries_the_table():
    """host only: the `gemm_tactics=` line of an engine header (Builder.build_engine writes it when a GPU is visible at build time)
    reaches the table when the session is created; a malformed line fails the creation with the parser's message."""
    from tensorrt_llm.runtime.native import NativeSession
    lib = _lib()
    lib.tllm_gemm_tactics_clear()
    cfg = dict(num_layers=1, num_heads=2, hidden_size=64, inter_size=24, vocab_size=128, quant_mode=0)
    s = NativeSession(dict(cfg, gemm_tactics='0:256:192:256:8:11.50;3:1024:12288:4096:20:45.10;'))
    assert lib.tllm_gemm_tactic_lookup(0, 256, 192, 256) == 8 and lib.tllm_gemm_tactic_lookup(3, 1024, 12288, 4096) == 20
    s.close()
    with pytest.raises(RuntimeError, match='parse'):
        NativeSession(dict(cfg, gemm_tactics='0:256:nonsense'))
    lib.tllm_gemm_tactics_clear()


@pytest.mark.gpu
def test_builder_profiles_on_the_device_and_the_engine_brings_the_table():
    """Builder._profile_gemm_tactics (what build_engine calls): the layer's four GEMM shapes at every power-of-two M up to
    max_batch_size * max_input_len, on the device; the text is what a session imports from the engine header."""
    from tensorrt_llm.builder import Builder
    lib = _lib()
    lib.tllm_gemm_tactics_clear()
    text = Builder._profile_gemm_tactics(dict(hidden_size=256, tensor_parallel=1, max_batch_size=2, max_input_len=48, quant_mode=0), 512)
    entries = [e.split(':') for e in text.split(';') if e]
    shapes = {(int(e[2]), int(e[3])) for e in entries}
    ms = {int(e[1]) for e in entries}
    assert shapes == {(768, 256), (256, 256), (512, 256), (256, 512)} and ms == {32, 64, 96}, text
    assert all(int(e[0]) == 0 and int(e[4]) > 0 and float(e[5]) > 0 for e in entries)
    lib.tllm_gemm_tactics_clear()
    assert lib.tllm_gemm_tactics_import(text.encode()) == 0
    assert lib.tllm_gemm_tactic_lookup(0, 96, 768, 256) > 0 and lib.tllm_gemm_tactic_lookup(0, 80, 768, 256) > 0  # bucket (64, 128]
    lib.tllm_gemm_tactics_clear()
