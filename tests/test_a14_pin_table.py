"""the step-dependent host tensors of the reference's decode loop
(T/tensorrt_llm/runtime/generation.py:490-770, :812-821, :852-946), pinned for B = 2, input_lengths = [3, 5], max_new = 4.

The fixture tests/golden/a14_host_step_table.json is produced by tests/golden/make_a14_table.py, a line-by-line restatement
of the reference's integer logic; the literals below are the same table derived by hand.  Pinned against it:
  * the oracle's calling convention for the generation attention (CPU);
  * the product, which keeps these tensors in DEVICE memory and advances them in the sampler kernel instead of rebuilding
    them on the host every step (GPU: tllm_session_get_step_state after the prompt and after every step)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def table():
    return json.load(open(os.path.join(GOLD, 'a14_host_step_table.json')))


def test_fixture_equals_the_hand_derived_table():
    t = table()
    assert (t['batch_size'], t['input_lengths'], t['max_input_length'], t['max_new_tokens'], t['max_seq_length']) == (2, [3, 5], 5, 4, 9)
    # masked_tokens[b, len_b:max_in] = 1   (generation.py:812-821)
    assert t['masked_tokens'] == [[0, 0, 0, 1, 1, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0, 0]]
    runs = t['runs']
    assert [r['phase'] for r in runs] == ['context', 'generation', 'generation', 'generation']
    # sequence_length = max_in + step, with the generation runs prepared one loop iteration early (:576-577, :686-687, :925-946)
    assert [r['sequence_length'] for r in runs] == [[5, 5], [5, 5], [6, 6], [7, 7]]
    # past_key_value_length = [0, is_context = 1], then [past_len, 0]   (:578-579, :688-689)
    assert [r['past_key_value_length'] for r in runs] == [[0, 1], [5, 0], [6, 0], [7, 0]]
    # position_ids: arange over the padded prompt, then input_lengths + step   (:735-750, :752-767)
    assert runs[0]['position_ids'] == [[0, 1, 2, 3, 4], [0, 1, 2, 3, 4]]
    assert [r['position_ids'] for r in runs[1:]] == [[[3], [5]], [[4], [6]], [[5], [7]]]
    # last_token_ids: the input lengths, then ones
    assert [r['last_token_ids'] for r in runs] == [[3, 5], [1, 1], [1, 1], [1, 1]]
    assert t['kv_slot_written'] == [None, 5, 6, 7]


def test_oracle_generation_steps_follow_the_table(monkeypatch):
    """oracle/quant_oracle.py::_forward must call the generation attention with the table's past length, padding mask and
    (through timestep - (max_input_len - input_len), MM/...Template.h:1425-1426) rotary positions."""
    from oracle import llama_oracle as O
    from oracle import quant_oracle as QO
    t = table()
    r = np.random.default_rng(0)
    L, H, D, I, V = 1, 2, 32, 48, 64
    w = {'vocab_embedding.weight': r.standard_normal((V, D)), 'ln_f.weight': np.ones(D), 'lm_head.weight': r.standard_normal((V, D)) * 0.1,
         'layers.0.input_layernorm.weight': np.ones(D), 'layers.0.post_layernorm.weight': np.ones(D),
         'layers.0.attention.qkv.weight': r.standard_normal((3 * D, D)) * 0.1, 'layers.0.attention.dense.weight': r.standard_normal((D, D)) * 0.1,
         'layers.0.mlp.fc.weight': r.standard_normal((I, D)) * 0.1, 'layers.0.mlp.gate.weight': r.standard_normal((I, D)) * 0.1,
         'layers.0.mlp.proj.weight': r.standard_normal((D, I)) * 0.1}
    w = {k: v.astype(np.float16) for k, v in w.items()}
    cfg = dict(num_layers=L, num_heads=H, hidden_size=D, inter_size=I, vocab_size=V)
    calls = []
    real = O.mmha_decode

    def spy(qkv16, cache, seq_len, input_lengths, max_input_len, timestep, num_heads, head_size, rot_dim, neox=True, q_scaling=1.0,
            masked_tokens=None, *a, **k):
        calls.append(dict(sequence_length=[int(x) for x in seq_len], past_len=int(timestep),
                          position=[int(timestep) - (int(max_input_len) - int(n)) for n in input_lengths],
                          masked=np.asarray(masked_tokens).tolist()))
        return real(qkv16, cache, seq_len, input_lengths, max_input_len, timestep, num_heads, head_size, rot_dim, neox, q_scaling,
                    masked_tokens, *a, **k)

    monkeypatch.setattr(O, 'mmha_decode', spy)
    B, S = t['batch_size'], t['max_input_length']
    lens = np.array(t['input_lengths'], np.int32)
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, V, lens[b])
    QO.run_fp16_model(cfg, w, ids, lens, t['max_new_tokens'])
    gen = [run for run in t['runs'] if run['phase'] == 'generation']
    assert len(calls) == len(gen) * L
    for c, run in zip(calls, gen):
        assert c['sequence_length'] == run['sequence_length']
        assert c['past_len'] == run['past_key_value_length'][0]
        assert c['position'] == [p[0] for p in run['position_ids']]
        assert c['masked'] == t['masked_tokens']


@pytest.mark.gpu
@pytest.mark.parametrize('use_graph', [False, True])
def test_device_resident_step_state_follows_the_table(use_graph):
    """The product never builds these tensors on the host after the prompt: the sampler kernel advances sequence_length and
    prepares the next rotary position on the device (one hipGraph replays every step).  After the prompt and after every
    step the device state must be what the reference would feed the NEXT run."""
    from tensorrt_llm.runtime.native import NativeSession
    from test_gpu_session import synth_model
    t = table()
    cfg, w = synth_model(7, L=2, H=2, D=64, I=96, V=128)
    s = NativeSession(dict(cfg, quant_mode=0))
    for k, v in w.items():
        s.set_tensor(k, v)
    s.finalize()
    B, S, NEW = t['batch_size'], t['max_input_length'], t['max_new_tokens']
    lens = np.array(t['input_lengths'], np.int32)
    r = np.random.default_rng(1)
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    s.setup(B, S, NEW)
    s.context(ids, lens)
    gen = [run for run in t['runs'] if run['phase'] == 'generation']
    for k, run in enumerate(gen):
        st = s.step_state()
        assert st['sequence_length'].tolist() == run['sequence_length'], f'before generation run {k + 1}'
        assert st['sequence_length'][0] == run['past_key_value_length'][0]
        assert st['next_position'].tolist() == [p[0] for p in run['position_ids']]
        assert st['masked_tokens'].tolist() == t['masked_tokens']
        assert st['input_lengths'].tolist() == t['input_lengths']
        s.step(1, use_graph=use_graph)
    out = s.output_ids()
    # the token run k consumed was written to slot kv_slot_written[k] of the id record, the prompt stays in place
    np.testing.assert_array_equal(out[:, :S], ids)
    assert out.shape == (B, t['max_seq_length'])
    s.close()
