"""summarize.py: the in-repo ROUGE (CPU) and the engine-vs-HF accuracy run on the GPU (SURVEY.md sections 3.3, 8d)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant')
sys.path.insert(0, EX)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def test_rouge_known_values():
    import summarize as S
    assert S.rouge_n('the cat sat on the mat', 'the cat sat on the mat', 1) == 1.0
    assert S.rouge_l('a b c', 'x y z') == 0.0
    # unigram overlap {the, cat}: P = 2/3, R = 2/4 -> F = 4/7
    assert abs(S.rouge_n('The cat sat.', 'the cat was sitting', 1) - 4 / 7) < 1e-12
    # bigrams: pred {the cat, cat sat}, ref {the cat, cat was, was sitting}: 1 match -> P = 1/2, R = 1/3 -> F = 0.4
    assert abs(S.rouge_n('the cat sat', 'the cat was sitting', 2) - 0.4) < 1e-12
    # LCS('a b c d e', 'a x c y e') = 3 -> P = R = 3/5
    assert abs(S.rouge_l('a b c d e', 'a x c y e') - 0.6) < 1e-12
    # clipped counts: 'the the the' vs 'the cat' -> 1 match, P = 1/3, R = 1/2 -> F = 0.4
    assert abs(S.rouge_n('the the the', 'the cat', 1) - 0.4) < 1e-12
    # rougeLsum on single-line texts equals rougeL; on two lines it is the union LCS
    assert abs(S.rouge_lsum('a b c d e', 'a x c y e') - 0.6) < 1e-12
    assert abs(S.rouge_lsum('a b\nc d', 'a b c d') - 1.0) < 1e-12
    m = S.Rouge()
    m.add_batch(['t1 t2 t3', 't4'], ['t1 t2 t3', 't5'])
    r = m.compute()
    assert r['rouge1'] == 50.0 and r['rougeL'] == 50.0 and set(r) == {'rouge1', 'rouge2', 'rougeL', 'rougeLsum'}


@pytest.mark.gpu
# observed (r03, deterministic kernels): fp16 1.000 / ROUGE-L 100, SmoothQuant + int8 KV 0.792 / 90.6 on this tiny two-layer model
@pytest.mark.parametrize('flags,min_match', [([], 0.95), (['--use_smooth_quant', '--per_channel', '--int8_kv_cache'], 0.65)])
def test_summarize_engine_vs_hf(tmp_path, flags, min_match):
    """hf_llama_convert -> build -> summarize.py --test_hf --test_trt_llm on seeded token prompts (ragged batch of 2):
    the fp16 engine reproduces HF's greedy continuation almost token for token; the SmoothQuant engine stays close."""
    import test_convert as T
    _, hf_dir = T.tiny_hf(tmp_path)
    ft = tmp_path / 'ft'
    subprocess.run([sys.executable, os.path.join(EX, 'hf_llama_convert.py'), '-i', hf_dir, '-o', str(ft), '-sq', '0.5',
                    '--calibrate-kv-cache', '--calib-samples', '8', '--calib-len', '64'], check=True, cwd=EX, timeout=600)
    eng = tmp_path / 'eng'
    subprocess.run([sys.executable, os.path.join(EX, 'build.py'), '--model_dir', str(ft / '1-gpu'), '--output_dir', str(eng),
                    '--max_batch_size', '2', '--max_input_len', '64', '--max_output_len', '16', '--log_level', 'error'] + flags,
                   check=True, cwd=EX, timeout=600)
    out = tmp_path / 'rouge.json'
    subprocess.run([sys.executable, os.path.join(EX, 'summarize.py'), '--hf_model_location', hf_dir, '--test_hf',
                    '--test_trt_llm', '--engine_dir', str(eng), '--synthetic', '--synthetic_len', '40', '--output_len', '12',
                    '--batch_size', '2', '--max_ite', '4', '--log_level', 'error', '--output_json', str(out)],
                   check=True, cwd=EX, timeout=900)
    r = json.load(open(out))
    print(f"[summarize {' '.join(flags) or 'fp16'}] token_match_rate {r['token_match_rate']:.3f}, rougeL vs HF {r['tensorrt_llm_vs_hf']['rougeL']:.1f}")
    assert r['token_match_rate'] >= min_match, r
    assert r['tensorrt_llm_vs_hf']['rougeL'] >= 100 * min_match, r
