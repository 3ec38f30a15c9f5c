"""bench_parity.FakeQuantSQ (the torch restatement of the SmoothQuant-static + int8-KV algorithm that bench.py's accuracy report
uses to split "engine <-> HF" into "kernels <-> algorithm" and "algorithm <-> HF") pinned to the numpy oracle
(oracle/quant_oracle.py, itself pinned to the reference's known-answer formulas): same tensors, same teacher-forced tokens,
context row + generation rows.  CPU only - a checker checking a checker."""
import numpy as np
import torch

import bench_parity
from oracle import quant_oracle as QO


def synth_model(seed, L=2, H=4, D=256, I=512, V=512):
    r = np.random.default_rng(seed)
    xav = lambda n, k: r.uniform(-1, 1, (n, k)) * np.sqrt(6.0 / (n + k)) * 2
    w = {'vocab_embedding.weight': r.standard_normal((V, D)) * 0.5, 'ln_f.weight': 1 + 0.1 * r.uniform(-1, 1, D),
         'lm_head.weight': xav(V, D)}
    for i in range(L):
        p = f'layers.{i}.'
        w[p + 'input_layernorm.weight'] = 1 + 0.1 * r.uniform(-1, 1, D)
        w[p + 'post_layernorm.weight'] = 1 + 0.1 * r.uniform(-1, 1, D)
        for n, (a, b) in {'attention.qkv': (3 * D, D), 'attention.dense': (D, D), 'mlp.fc': (I, D), 'mlp.gate': (I, D),
                          'mlp.proj': (D, I)}.items():
            w[p + n + '.weight'] = xav(a, b)
    w = {k: v.astype(np.float16) for k, v in w.items()}
    return dict(num_layers=L, num_heads=H, hidden_size=D, inter_size=I, vocab_size=V, max_position_embeddings=128,
                rms_norm_eps=1e-6), w


def test_fakequant_restatement_equals_the_oracle_on_a_small_model():
    cfg, w = synth_model(11)
    S, NEW = 12, 5
    r = np.random.default_rng(5)
    ids = r.integers(3, cfg['vocab_size'], (2, S)).astype(np.int32)
    lens = np.array([S, S], np.int32)
    qmodel = QO.quantise_model(cfg, w, 'sq_static_pc', 1, calib_ids=ids, calib_lens=lens)
    feed = r.integers(3, cfg['vocab_size'], (2, NEW)).astype(np.int32)  # teacher-forced continuation
    taps = {}
    ref_logits, _ = QO.run_model(qmodel, ids, lens, NEW, feed_ids=feed[:, 1:], taps=taps)
    # the oracle consumed ids, then feed[:, 1], feed[:, 2], ...: logits[k] belongs to the prefix ids + feed[:, 1:k+1]
    tensors = {k: torch.from_numpy(np.array(v)) for k, v in qmodel['engine_tensors'].items()}
    fq = bench_parity.FakeQuantSQ(torch, tensors, cfg['num_layers'], heads=cfg['num_heads'])
    full = torch.from_numpy(np.concatenate([ids, feed[:, 1:NEW]], axis=1).astype(np.int64))
    ftaps = []
    got = fq.forward(full, S, taps=ftaps, first_row=S - 1).numpy()  # [2, NEW, V]
    ref = np.stack(ref_logits, 1)  # [2, NEW, V]
    scale = np.abs(ref).max()
    # identical integers at every quantiser of every generation row -> the logits agree to fp32 summation noise; a 1-LSB flip
    # (an fp16 tie broken differently by the float64 / float32 reductions) would show up as ~1e-2 of the scale
    for li in range(cfg['num_layers']):
        for name in ('qkv_in', 'o_in', 'mlp_in', 'proj_in'):
            for step in range(NEW - 1):
                o = taps['gemm_in'][step][li][name].astype(np.int32)  # [B, width]
                f = ftaps[li][name].reshape(2, full.shape[1], -1)[:, S + step].numpy().astype(np.int32)
                d = np.abs(o - f)
                assert d.max() <= 1 and (d != 0).mean() < 0.01, (li, name, step, d.max(), (d != 0).mean())
    assert np.abs(got - ref).max() < 2e-2 * scale
    assert np.abs(got - ref).mean() < 2e-3 * scale
