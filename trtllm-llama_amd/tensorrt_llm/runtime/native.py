"""ctypes binding of include/tllm_runtime_api.h (the C++ host decode loop).

The reference's GenerationSession drives a TensorRT execution context from Python
(T/tensorrt_llm/runtime/generation.py:43-100); here the same object drives `tllm_session_*`.
torch tensors are handed over as raw device pointers (tensor handoff only)."""
import ctypes
from typing import Dict, Optional, Sequence

import numpy as np

from ..plugin import capi

_NP2C = {np.dtype(np.float32): capi.FLOAT, np.dtype(np.float16): capi.HALF, np.dtype(np.int8): capi.INT8,
         np.dtype(np.int32): capi.INT32, np.dtype(np.uint8): capi.INT8}

_bound = False


def _lib():
    global _bound
    lib = capi.load_library()
    if _bound:
        return lib
    c = ctypes
    lib.tllm_session_create.argtypes = [c.c_char_p]
    lib.tllm_session_create.restype = c.c_void_p
    lib.tllm_session_set_tensor.argtypes = [c.c_void_p, c.c_char_p, c.c_int32, c.POINTER(c.c_int64), c.c_int32,
                                            c.c_void_p, c.c_int32]
    lib.tllm_session_set_tensor.restype = c.c_int32
    lib.tllm_session_finalize.argtypes = [c.c_void_p]
    lib.tllm_session_finalize.restype = c.c_int32
    lib.tllm_session_load_engine.argtypes = [c.c_void_p, c.c_size_t]
    lib.tllm_session_load_engine.restype = c.c_void_p
    lib.tllm_session_setup.argtypes = [c.c_void_p, c.c_int32, c.c_int32, c.c_int32]
    lib.tllm_session_setup.restype = c.c_int32
    lib.tllm_session_setup_beam.argtypes = [c.c_void_p, c.c_int32, c.c_int32, c.c_int32, c.c_int32]
    lib.tllm_session_setup_beam.restype = c.c_int32
    lib.tllm_session_get_beam_output.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.tllm_session_get_beam_output.restype = c.c_int32
    lib.tllm_session_get_beam_state.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.tllm_session_get_beam_state.restype = c.c_int32
    lib.tllm_session_logit_rows.argtypes = [c.c_void_p]
    lib.tllm_session_logit_rows.restype = c.c_int32
    lib.tllm_session_vocab_size.argtypes = [c.c_void_p]
    lib.tllm_session_vocab_size.restype = c.c_int32
    lib.tllm_session_generate.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_int32, c.c_int32, c.c_int32,
                                          c.c_void_p, c.c_void_p]
    lib.tllm_session_generate.restype = c.c_int32
    lib.tllm_session_context.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.tllm_session_context.restype = c.c_int32
    lib.tllm_session_step.argtypes = [c.c_void_p, c.c_int32, c.c_int32, c.c_void_p]
    lib.tllm_session_step.restype = c.c_int32
    lib.tllm_session_fake_context.argtypes = [c.c_void_p, c.c_int32, c.c_uint32, c.c_void_p]
    lib.tllm_session_fake_context.restype = c.c_int32
    lib.tllm_session_get_logits.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p]
    lib.tllm_session_get_logits.restype = c.c_int32
    lib.tllm_session_get_output_ids.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p]
    lib.tllm_session_get_output_ids.restype = c.c_int32
    lib.tllm_session_kv_cache_ptr.argtypes = [c.c_void_p, c.c_int32]
    lib.tllm_session_kv_cache_ptr.restype = c.c_void_p
    lib.tllm_session_get_step_state.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p, c.c_void_p]
    lib.tllm_session_get_step_state.restype = c.c_int32
    lib.tllm_session_get_tap.argtypes = [c.c_void_p, c.c_int32, c.c_void_p, c.c_size_t, c.c_void_p]
    lib.tllm_session_get_tap.restype = c.c_int32
    lib.tllm_session_get_tap_ex.argtypes = [c.c_void_p, c.c_int32, c.c_int32, c.c_void_p, c.c_size_t, c.c_void_p]
    lib.tllm_session_get_tap_ex.restype = c.c_int32
    lib.tllm_session_force_tokens.argtypes = [c.c_void_p, c.c_void_p, c.c_void_p]
    lib.tllm_session_force_tokens.restype = c.c_int32
    lib.tllm_session_step_bytes.argtypes = [c.c_void_p, c.c_int32]
    lib.tllm_session_step_bytes.restype = c.c_int64
    lib.tllm_session_profile.argtypes = [c.c_void_p, c.c_int32, c.POINTER(c.c_float), c.POINTER(c.c_int64), c.c_void_p]
    lib.tllm_session_profile.restype = c.c_int32
    lib.tllm_session_time_kernel.argtypes = [c.c_void_p, c.c_int32, c.c_int32, c.POINTER(c.c_float), c.POINTER(c.c_int64),
                                             c.c_void_p]
    lib.tllm_session_time_kernel.restype = c.c_int32
    lib.tllm_session_fused_retries.argtypes = [c.c_void_p]
    lib.tllm_session_fused_retries.restype = c.c_int32
    lib.tllm_session_decode_form.argtypes = [c.c_void_p]
    lib.tllm_session_decode_form.restype = c.c_int32
    lib.tllm_session_destroy.argtypes = [c.c_void_p]
    lib.tllm_session_destroy.restype = None
    _bound = True
    return lib


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f'{what} failed: {capi.last_error()}')


class NativeSession:
    """Thin owner of a tllm_session_t."""

    def __init__(self, config: Optional[Dict] = None, engine: Optional[bytes] = None):
        lib = _lib()
        self._keep = []  # torch tensors whose device memory the session references
        if engine is not None:
            buf = (ctypes.c_char * len(engine)).from_buffer_copy(engine)
            self._h = lib.tllm_session_load_engine(buf, len(engine))
        else:
            text = '\n'.join(f'{k}={v}' for k, v in config.items())
            self._h = lib.tllm_session_create(text.encode())
        if not self._h:
            raise RuntimeError(f'tllm_session create failed: {capi.last_error()}')
        self.batch = self.max_in = self.max_new = 0
        self.vocab = int(config['vocab_size']) if config else (int(lib.tllm_session_vocab_size(self._h)) or None)

    def set_tensor(self, name: str, t):
        lib = _lib()
        if isinstance(t, np.ndarray):
            a = np.ascontiguousarray(t)
            dims = (ctypes.c_int64 * max(a.ndim, 1))(*a.shape)
            _check(lib.tllm_session_set_tensor(self._h, name.encode(), _NP2C[a.dtype], dims, a.ndim, a.ctypes.data, 0),
                   f'set_tensor({name})')
        else:  # torch cuda tensor: hand the device pointer over, keep the tensor alive
            import torch
            assert t.is_cuda and t.is_contiguous()
            code = {torch.float32: capi.FLOAT, torch.float16: capi.HALF, torch.int8: capi.INT8, torch.uint8: capi.INT8,
                    torch.int32: capi.INT32}[t.dtype]
            dims = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            _check(lib.tllm_session_set_tensor(self._h, name.encode(), code, dims, t.dim(), t.data_ptr(), 1),
                   f'set_tensor({name})')
            self._keep.append(t)

    def finalize(self):
        _check(_lib().tllm_session_finalize(self._h), 'finalize')

    def setup(self, batch: int, max_input_len: int, max_new_tokens: int, beam_width: int = 1):
        """batch prompts x beam_width hypotheses (beam search when > 1, beam_width <= 8; beyond 8 sequences the generation GEMVs run in slabs of 8 rows)."""
        _check(_lib().tllm_session_setup_beam(self._h, batch, beam_width, max_input_len, max_new_tokens), 'setup')
        self.batch, self.max_in, self.max_new, self.beam = batch, max_input_len, max_new_tokens, beam_width

    @staticmethod
    def _i32(a) -> np.ndarray:
        return np.ascontiguousarray(np.asarray(a), dtype=np.int32)

    def context(self, input_ids, input_lengths, stream: int = 0):
        ids, lens = self._i32(input_ids), self._i32(input_lengths)
        assert ids.shape == (self.batch, self.max_in) and lens.shape == (self.batch, )
        _check(_lib().tllm_session_context(self._h, ids.ctypes.data, lens.ctypes.data, stream), 'context')

    def step(self, n_steps: int = 1, use_graph: bool = False, stream: int = 0):
        _check(_lib().tllm_session_step(self._h, n_steps, 1 if use_graph else 0, stream), 'step')

    def fake_context(self, length: int, seed: int = 0, stream: int = 0):
        _check(_lib().tllm_session_fake_context(self._h, length, seed, stream), 'fake_context')

    def logits(self, vocab: Optional[int] = None, stream: int = 0) -> np.ndarray:
        v = vocab or self.vocab
        rows = _lib().tllm_session_logit_rows(self._h)  # batch after the prompt, batch * beam after a step
        out = np.empty((rows, v), np.float32)
        _check(_lib().tllm_session_get_logits(self._h, out.ctypes.data, stream), 'get_logits')
        return out

    def output_ids(self, stream: int = 0) -> np.ndarray:
        """Per-sequence token record [batch * beam, max_seq_len] (with beam search: the un-back-tracked step ids)."""
        out = np.empty((self.batch * self.beam, self.max_in + self.max_new), np.int32)
        _check(_lib().tllm_session_get_output_ids(self._h, out.ctypes.data, stream), 'get_output_ids')
        return out

    def beam_output(self, stream: int = 0):
        """(ids [batch, beam, max_seq_len] back-tracked and best first, cum_log_probs [batch, beam])."""
        out = np.empty((self.batch, self.beam, self.max_in + self.max_new), np.int32)
        cum = np.empty((self.batch, self.beam), np.float32)
        _check(_lib().tllm_session_get_beam_output(self._h, out.ctypes.data, cum.ctypes.data if self.beam > 1 else None, stream),
               'get_beam_output')
        return out, (cum if self.beam > 1 else None)

    def beam_state(self, stream: int = 0):
        """dict(parent_ids, cache_indirection [batch * beam, max_seq_len], finished, sequence_lengths [batch * beam])."""
        n, smax = self.batch * self.beam, self.max_in + self.max_new
        d = dict(parent_ids=np.empty((n, smax), np.int32), cache_indirection=np.empty((n, smax), np.int32),
                 finished=np.empty(n, np.int32), sequence_lengths=np.empty(n, np.int32))
        _check(_lib().tllm_session_get_beam_state(self._h, d['parent_ids'].ctypes.data, d['cache_indirection'].ctypes.data,
                                                  d['finished'].ctypes.data, d['sequence_lengths'].ctypes.data, stream),
               'get_beam_state')
        return d

    def generate(self, input_ids, input_lengths, max_new_tokens: int, end_id: int = -1, pad_id: int = 0,
                 stream: int = 0) -> np.ndarray:
        ids, lens = self._i32(input_ids), self._i32(input_lengths)
        shape = (self.batch, self.max_in + self.max_new) if self.beam == 1 else (self.batch, self.beam, self.max_in + self.max_new)
        out = np.empty(shape, np.int32)
        _check(_lib().tllm_session_generate(self._h, ids.ctypes.data, lens.ctypes.data, max_new_tokens, end_id, pad_id,
                                            out.ctypes.data, stream), 'generate')
        return out

    PROFILE_CLASSES = ('gemv_layer', 'gemv_head', 'attention', 'other', 'comm')

    def profile(self, n_steps: int, stream: int = 0):
        """{class: (total ms, launches)} over n_steps instrumented (eager) generation steps."""
        ms = (ctypes.c_float * 5)()
        cnt = (ctypes.c_int64 * 5)()
        _check(_lib().tllm_session_profile(self._h, n_steps, ms, cnt, stream), 'profile')
        return {n: (float(ms[i]), int(cnt[i])) for i, n in enumerate(self.PROFILE_CLASSES)}

    LAYER_KERNELS = {'qkv': 1, 'attention': 2, 'o_proj': 4, 'gate_up': 5, 'down': 6, 'front': 7, 'mlp': 8}

    def fused_retries(self) -> int:
        """requests generate() repeated behind an expired in-launch wait of the one-launch projection + attention"""
        return int(_lib().tllm_session_fused_retries(self._h))

    def decode_form(self) -> int:
        """bit 0: QKV projection + attention in one launch, bit 1: + the O-projection stage, bit 2: the gated MLP in one launch"""
        return int(_lib().tllm_session_decode_form(self._h))

    def time_kernel(self, which: str, sweeps: int = 4, stream: int = 0):
        """(average microseconds per launch, launches) of one per-layer kernel launched back to back over all layers."""
        us = ctypes.c_float()
        n = ctypes.c_int64()
        _check(_lib().tllm_session_time_kernel(self._h, self.LAYER_KERNELS[which], sweeps, ctypes.byref(us),
                                               ctypes.byref(n), stream), 'time_kernel')
        return float(us.value), int(n.value)

    def kv_cache_ptr(self, layer: int) -> int:
        return _lib().tllm_session_kv_cache_ptr(self._h, layer)

    def step_state(self, stream: int = 0):
        """dict(sequence_length, next_position, input_lengths [batch * beam], masked_tokens [batch * beam, max_seq_len]): the
        device-resident counterparts of the per-step host tensors of the reference's decode loop."""
        n, smax = self.batch * self.beam, self.max_in + self.max_new
        d = dict(sequence_length=np.empty(n, np.int32), next_position=np.empty(n, np.int32),
                 masked_tokens=np.empty((n, smax), np.int32), input_lengths=np.empty(n, np.int32))
        _check(_lib().tllm_session_get_step_state(self._h, d['sequence_length'].ctypes.data, d['next_position'].ctypes.data,
                                                  d['masked_tokens'].ctypes.data, d['input_lengths'].ctypes.data, stream),
               'get_step_state')
        return d

    def force_tokens(self, ids, stream: int = 0):
        """teacher forcing: overwrite the token the last step chose (parity tests compare on identical prefixes)"""
        ids = self._i32(ids)
        assert ids.shape == (self.batch, )
        _check(_lib().tllm_session_force_tokens(self._h, ids.ctypes.data, stream), 'force_tokens')

    def attention_tap(self, layer: int, heads_x_dh: int, quantised: bool, stream: int = 0) -> np.ndarray:
        """O-projection input of the last generation step (debug_taps=1): [batch * beam, H/tp * Dh] fp16, int8 for SmoothQuant."""
        out = np.empty((self.batch * self.beam, heads_x_dh), np.int8 if quantised else np.float16)
        _check(_lib().tllm_session_get_tap(self._h, layer, out.ctypes.data, out.nbytes, stream), 'get_tap')
        return out

    TAPS = {'qkv_in': 0, 'o_in': 1, 'mlp_in': 2, 'proj_in': 3, 'x_in': 4}

    def tap(self, layer: int, which: str, width: int, quantised: bool, stream: int = 0) -> np.ndarray:
        """Input of one of the layer's four GEMMs in the last generation step, behind its prologue (debug_taps=1):
        [batch * beam, width] fp16, int8 for SmoothQuant (the output of that GEMM's activation quantiser)."""
        out = np.empty((self.batch * self.beam, width), np.int8 if (quantised and which != 'x_in') else np.float16)
        _check(_lib().tllm_session_get_tap_ex(self._h, layer, self.TAPS[which], out.ctypes.data, out.nbytes, stream), 'get_tap_ex')
        return out

    def step_bytes(self, context_len: int) -> int:
        return _lib().tllm_session_step_bytes(self._h, context_len)

    def close(self):
        if getattr(self, '_h', None):
            _lib().tllm_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
