"""The tensorrt_llm-shaped Python front-end: tracing, module surgery, engine build (CPU), and — on the GPU — the
build.py -> run.py flow against the HF golden values."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant')
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, EX)

TINY = ['--n_layer', '2', '--n_head', '2', '--n_embd', '64', '--inter_size', '96', '--vocab_size', '128',
        '--n_positions', '64', '--max_batch_size', '2', '--max_input_len', '16', '--max_output_len', '8']


def hf_state_dict(t):
    """golden npz (reference module naming) -> HF state-dict naming, to exercise load_from_hf_llama."""
    sd = {'model.embed_tokens.weight': t['vocab_embedding.weight'], 'model.norm.weight': t['ln_f.weight'],
          'lm_head.weight': t['lm_head.weight']}
    for i in range(2):
        p, q = f'layers.{i}.', f'model.layers.{i}.'
        qkv = t[p + 'attention.qkv.weight']
        d = qkv.shape[1]
        for j, n in enumerate('qkv'):
            sd[q + f'self_attn.{n}_proj.weight'] = qkv[j * d:(j + 1) * d]
        sd[q + 'self_attn.o_proj.weight'] = t[p + 'attention.dense.weight']
        sd[q + 'input_layernorm.weight'] = t[p + 'input_layernorm.weight']
        sd[q + 'post_attention_layernorm.weight'] = t[p + 'post_layernorm.weight']
        sd[q + 'mlp.gate_proj.weight'] = t[p + 'mlp.fc.weight']
        sd[q + 'mlp.up_proj.weight'] = t[p + 'mlp.gate.weight']
        sd[q + 'mlp.down_proj.weight'] = t[p + 'mlp.proj.weight']
    return sd


def build_tiny_engine(quant_mode=None, tp=1, rank=0):
    import tensorrt_llm
    from tensorrt_llm.models import LLaMAForCausalLM, weight_only_quantize
    from tensorrt_llm.network import net_guard
    from tensorrt_llm.quantization import QuantMode
    from weight import load_from_hf_llama
    qm = quant_mode if quant_mode is not None else QuantMode(0)
    t = dict(np.load(os.path.join(GOLD, 'hf_tiny_llama.npz')))
    model = LLaMAForCausalLM(num_layers=2, num_heads=2, hidden_size=64, vocab_size=128, hidden_act='silu',
                             max_position_embeddings=64, dtype='float16', mlp_hidden_size=24, tensor_parallel=tp,
                             tensor_parallel_group=list(range(tp)), quant_mode=qm)
    if qm.is_weight_only():
        model = weight_only_quantize(model, qm)
    load_from_hf_llama(model, hf_state_dict(t), rank, tp, 'float16')
    builder = tensorrt_llm.Builder()
    net = builder.create_network()
    net.plugin_config.set_gpt_attention_plugin('float16')
    net.plugin_config.set_gemm_plugin('float16')
    if qm.is_weight_only():
        net.plugin_config.set_weight_only_quant_matmul_plugin('float16')
    if tp > 1:
        net.plugin_config.set_nccl_plugin('float16')
    with net_guard(net):
        net.set_named_parameters(model.named_parameters())
        model(*model.prepare_inputs(2, 16, 8, True, 1))
    cfg = builder.create_builder_config(name='llama', precision='float16', tensor_parallel=tp, num_layers=2, num_heads=2,
                                        hidden_size=64, vocab_size=128, hidden_act='silu', max_position_embeddings=64,
                                        inter_size=24, quant_mode=int(qm), tp_rank=rank)
    engine = builder.build_engine(net, cfg)
    assert engine is not None
    return engine, net, t


def test_trace_records_reference_plugin_sequence():
    """One decoder layer must trace to the reference's node sequence: rms_norm -> Gemm(qkv) ->
    GPTAttention -> Gemm(dense) -> add -> rms_norm -> Gemm(fc) -> silu -> Gemm(gate) -> mul -> Gemm(proj) -> add."""
    _, net, _ = build_tiny_engine()
    seq = [n['attrs'].get('plugin_type', n['op']) for n in net.nodes
           if n['op'] not in ('constant', 'shape', 'assertion', 'mark_output')]
    layer = ['rms_norm', 'Gemm', 'GPTAttention', 'Gemm', 'add', 'rms_norm', 'Gemm', 'silu', 'Gemm', 'mul', 'Gemm', 'add']
    assert seq == ['embedding'] + layer * 2 + ['rms_norm', 'gather_last_token_logits', 'Gemm']
    attn = [n for n in net.nodes if n['attrs'].get('plugin_type') == 'GPTAttention'][0]
    f = attn['attrs']['fields']
    # field names / values of T/tensorrt_llm/functional.py:2833-2891
    assert list(f) == ['num_heads', 'head_size', 'unidirectional', 'q_scaling', 'rotary_embedding_dim', 'neox_rotary_style',
                       'context_fmha_type', 'multi_block_mode', 'multi_query_mode', 'int8_kv_cache', 'fp8_kv_cache',
                       'remove_input_padding', 'mask_type', 'paged_kv_cache', 'type_id', 'in_flight_batching']
    v = lambda k: np.ravel(f[k])[0]
    assert v('num_heads') == 2 and v('head_size') == 32 and v('rotary_embedding_dim') == 32 and v('q_scaling') == 1.0
    assert v('neox_rotary_style') == 1 and v('type_id') == 1 and v('mask_type') == 1
    assert len(attn['inputs']) == 8  # tensor, past_kv, sequence_length, past_kv_length, masked_tokens, input_lengths, max_input_length, cache_indirection
    names = [t.name for t in net.get_inputs()]
    for want in ['input_ids', 'position_ids', 'past_key_value_0', 'past_key_value_1', 'sequence_length',
                 'past_key_value_length', 'masked_tokens', 'input_lengths', 'max_input_length', 'last_token_ids',
                 'cache_indirection']:
        assert want in names
    assert list(net._outputs) == ['logits', 'present_key_value_0', 'present_key_value_1']


def test_int8_kv_adds_scale_inputs_and_tp_splits_heads():
    from tensorrt_llm.quantization import QuantMode
    import tensorrt_llm
    from tensorrt_llm.models import LLaMAForCausalLM
    from tensorrt_llm.network import net_guard
    m = LLaMAForCausalLM(num_layers=1, num_heads=4, hidden_size=128, vocab_size=131, hidden_act='silu',
                         max_position_embeddings=64, dtype='float16', mlp_hidden_size=64, tensor_parallel=2,
                         tensor_parallel_group=[0, 1], quant_mode=QuantMode(0).set_int8_kv_cache())
    assert m.lm_head.weight.shape == (66, 128)  # vocab padded to a multiple of tp, then split
    assert m.layers[0].attention.qkv.weight.shape == (3 * 64, 128)
    assert m.layers[0].attention.dense.weight.shape == (128, 64)
    assert m.layers[0].mlp.fc.weight.shape == (32, 128) and m.layers[0].mlp.proj.weight.shape == (128, 32)
    net = tensorrt_llm.Builder().create_network()
    net.plugin_config.set_gpt_attention_plugin('float16')
    net.plugin_config.set_gemm_plugin('float16')
    net.plugin_config.set_nccl_plugin('float16')
    with net_guard(net):
        m(*m.prepare_inputs(2, 16, 8, True, 1))
    attn = [n for n in net.nodes if n['attrs'].get('plugin_type') == 'GPTAttention'][0]
    assert len(attn['inputs']) == 10 and np.ravel(attn['attrs']['fields']['int8_kv_cache'])[0] == 1
    assert np.ravel(attn['attrs']['fields']['num_heads'])[0] == 2  # heads per rank
    plugins = [n['attrs'].get('plugin_type') for n in net.nodes if n['op'] == 'plugin']
    assert plugins.count('AllReduce') == 2 and plugins.count('AllGather') == 1  # 2 per layer + lm_head gather
    kv = [t for t in net.get_inputs() if t.name == 'past_key_value_0'][0]
    assert int(kv.dtype) == 2 and kv.shape == (-1, 2, 2, -1, 32)  # int8 cache, heads/tp


def test_build_cli_writes_engine_and_config(tmp_path):
    out = tmp_path / 'eng'
    cmd = [sys.executable, os.path.join(EX, 'build.py'), '--output_dir', str(out), '--use_weight_only', '--int8_kv_cache',
           '--log_level', 'error'] + TINY
    subprocess.run(cmd, check=True, cwd=EX, timeout=300)
    cfg = json.load(open(out / 'config.json'))
    bc = cfg['builder_config']
    for k in ['name', 'precision', 'tensor_parallel', 'num_layers', 'num_heads', 'hidden_size', 'vocab_size', 'hidden_act',
              'max_position_embeddings', 'max_batch_size', 'max_input_len', 'max_output_len', 'multi_query_mode']:
        assert k in bc, k  # T/tensorrt_llm/builder.py:138-142
    assert bc['quant_mode'] == 2 | 32 and bc['int8'] is True
    assert cfg['plugin_config']['gpt_attention_plugin'] == 'float16'
    assert cfg['plugin_config']['weight_only_quant_matmul_plugin'] == 'float16'
    eng = open(out / 'llama_float16_tp1_rank0.engine', 'rb').read()
    assert eng[:8] == b'TLLMENG1'


def test_paged_kv_cache_trace_adds_pool_and_block_pointer_inputs(tmp_path):
    """--paged_kv_cache: the GPTAttention node carries paged_kv_cache = 1 and one more input (the block pointers), the
    cache input becomes the block pool [blocks, 2, heads, tokens_per_block, head_size] (gptAttentionPlugin.cpp:206-253)."""
    out = tmp_path / 'eng'
    subprocess.run([sys.executable, os.path.join(EX, 'build.py'), '--output_dir', str(out), '--paged_kv_cache', '--int8_kv_cache',
                    '--log_level', 'error'] + TINY, check=True, cwd=EX, timeout=300)
    blob = open(out / 'llama_float16_tp1_rank0.engine', 'rb').read()
    assert b'paged_kv_cache=1' in blob[:4096] and b'tokens_per_block=64' in blob[:4096]
    assert b'kv_cache_block_pointers_0' in blob and b'kv_cache_block_pointers_1' in blob


def test_smooth_quant_and_weight_only_are_exclusive():
    from build import parse_arguments
    with pytest.raises(AssertionError):
        parse_arguments(['--use_smooth_quant', '--use_weight_only'])
    a = parse_arguments(['--use_smooth_quant', '--per_channel', '--int8_kv_cache'])
    assert int(a.quant_mode) == 46


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['fp16', 'woq8'])
def test_generation_session_matches_hf_golden(mode):
    from tensorrt_llm import Mapping
    from tensorrt_llm.quantization import QuantMode
    from tensorrt_llm.runtime import GenerationSession, ModelConfig, SamplingConfig
    qm = QuantMode(0) if mode == 'fp16' else QuantMode.use_weight_only()
    engine, _, t = build_tiny_engine(qm)
    sess = GenerationSession(ModelConfig(vocab_size=128, num_layers=2, num_heads=2, hidden_size=64), engine, Mapping(1, 0))
    ids, lens = t['ids'], t['input_lengths']
    B, S = ids.shape
    sess.setup(B, S, 6)
    out = sess.decode(ids, lens, SamplingConfig(end_id=-1, pad_id=2))
    assert out.shape == (B, 1, S + 6)
    np.testing.assert_array_equal(out[:, 0, :S], ids)
    if mode == 'fp16':
        np.testing.assert_array_equal(out[:, 0, S], t['next_ids'])
    logits = sess.runtime.logits()
    assert np.isfinite(logits).all()


@pytest.mark.gpu
@pytest.mark.parametrize('extra', [[], ['--remove_input_padding'], ['--paged_kv_cache']])
def test_build_then_run_cli(tmp_path, extra):
    out = tmp_path / 'eng'
    subprocess.run([sys.executable, os.path.join(EX, 'build.py'), '--output_dir', str(out), '--log_level', 'error'] + TINY
                   + extra, check=True, cwd=EX, timeout=300)
    if extra:
        blob = open(out / 'llama_float16_tp1_rank0.engine', 'rb').read()
        assert extra[0][2:].encode() + b'=1' in blob[:4096] and b'"input_ids"' in blob
    np.save(tmp_path / 'in.npy', np.array([5, 17, 99, 3, 64], np.int32))
    r = subprocess.run([sys.executable, os.path.join(EX, 'run.py'), '--max_output_len', '8', '--engine_dir', str(out),
                        '--input_tokens', str(tmp_path / 'in.npy'), '--output_npy', str(tmp_path / 'out.npy'),
                        '--num_runs', '7'], check=True, cwd=EX, timeout=600, capture_output=True, text=True)
    assert 'llama-run (mean latency:' in r.stdout
    o = np.load(tmp_path / 'out.npy')
    assert o.shape == (1, 13) and list(o[0, :5]) == [5, 17, 99, 3, 64]
