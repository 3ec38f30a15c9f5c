"""The accuracy half of the metric on a parent where it can go EITHER way (the deterministic trained parent saturates
the instrument - HF ROUGE-L 100.0, margins >= 5.5, delta 0.0 for an engine with error 0.04 and for one with error 2.9).

Parent: tests/golden/trained_llama_stochastic (tests/golden/train_stochastic_llama.py): the same small LLaMA trained on the
"records and phrases" language made stochastic - 2 - 4 near-equiprobable successors for half of the phrases, two near-equiprobable
tokens at 40 % of the body positions.  HF fp32 on it: ROUGE-L 43.7 against the references (one SAMPLED continuation each, as a
human summary is one draw), top-1 / top-2 margin below 1.0 on 29 % of its steps and below 0.2 on 16 % - the regime of the
reference's own table (HF 15.2, its fp16 engine 1.5 away: T/README.md:912-921).

Procedure = the reference's (T/examples/llama_quant/summarize.py:91,260,321-323,352), through the product's command lines
    hf_llama_convert.py -> build.py <flags> -> summarize.py --test_hf --test_trt_llm
on 256 prompts x 100 new tokens (the reference: 20 x 100), HF's continuations taken from the fixture (HF fp32 on the CPU, computed
once).  A flipped near-tie sends a whole continuation elsewhere, so the delta of a finite sample carries noise: summarize.py reports
it with a paired bootstrap interval over the prompts, and the criterion is read on the interval -

    PASS  iff  the 95 % interval of (ROUGE-L engine - ROUGE-L HF) intersects [-1, +1]   ("within about 1", README.md:921)

- i.e. the sample does not show the engine more than one point away from HF - AND, for BASELINE.json's configurations, the point
estimate itself within +-1.  A mildly broken engine (SmoothQuant static calibrated with every activation range a third of the true
one) must FAIL: its whole interval lies below -1.
"""
import concurrent.futures
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

import trained_parents as TP

ROOT = TP.ROOT
EX = os.path.join(ROOT, 'trtllm-llama_amd', 'examples', 'llama_quant')
FIX = TP.PARENTS['stochastic']
NEW = 100
N_PROMPTS = 256
BATCH = 8

CONFIGS = {
    'fp16': (False, []),
    'int8_kv': (False, ['--int8_kv_cache']),
    'woq8_int8kv': (False, ['--use_weight_only', '--int8_kv_cache']),
    'woq4_int8kv': (False, ['--use_weight_only', '--weight_only_precision', 'int4', '--int8_kv_cache']),
    'sq_static_int8kv': (True, ['--use_smooth_quant', '--per_channel', '--int8_kv_cache']),
    'sq_per_token_int8kv': (True, ['--use_smooth_quant', '--per_token', '--per_channel', '--int8_kv_cache']),
    # not a reference flag: the down_proj input - the heavy-tailed SwiGLU product - smoothed with alpha 1.0 (--smoothquant-down).
    # It halves the mean logit error on the deterministic parent; HERE it costs token matches (0.876 -> 0.815): measured, not a default
    'sq_static_int8kv_down1': ('down1', ['--use_smooth_quant', '--per_channel', '--int8_kv_cache']),
}


def test_fixture_is_a_sensitive_parent():
    """What makes the criterion decidable in both directions, checked on the committed fixture (CPU)."""
    info = json.load(open(os.path.join(FIX, 'TRAINLOG.json')))
    e = TP.load_eval('stochastic')
    assert e['prompts'].shape[0] >= 200 and e['hf_tokens'].shape[1] == NEW
    assert 30.0 <= info['hf_rougeL_vs_reference']['mean'] <= 70.0, info['hf_rougeL_vs_reference']
    assert info['margin']['frac_below_1p0'] >= 0.20, info['margin']
    assert info['language']['stochastic_position_fraction'] >= 0.25
    assert info['distinct_tokens_per_continuation'] > 40  # no cycles
    m = e['hf_margins']
    assert abs(float(np.mean(m < 1.0)) - info['margin']['frac_below_1p0']) < 1e-6
    lg = e['hf_logits'].astype(np.float32)
    np.testing.assert_array_equal(lg.argmax(-1), e['hf_tokens'][:lg.shape[0]])


@pytest.fixture(scope='module')
def ft_dirs(tmp_path_factory):
    base = tmp_path_factory.mktemp('stochastic')
    e = TP.load_eval('stochastic')
    calib = base / 'calib.npy'
    np.save(calib, e['calib'])
    out, cmds = {}, {}
    for sq in (False, True, 'down1'):
        d = base / {False: 'ft', True: 'ft_sq', 'down1': 'ft_sq_down1'}[sq]
        cmds[sq] = [sys.executable, os.path.join(EX, 'hf_llama_convert.py'), '-i', FIX, '-o', str(d), '--calibrate-kv-cache',
                    '--calib-ids', str(calib)] + (['-sq', '0.5'] if sq else []) + (['--smoothquant-down', '1.0'] if sq == 'down1' else [])
        out[sq] = str(d / '1-gpu')
    # (the three conversions side by side: they are independent processes, and the suite's wall time is a budget - VERDICT r05)
    with concurrent.futures.ThreadPoolExecutor(3) as ex:
        for r in ex.map(lambda c: subprocess.run(c, cwd=EX, timeout=900, capture_output=True, text=True), cmds.values()):
            assert r.returncode == 0, r.stderr[-3000:]
    np.save(base / 'prompts.npy', e['prompts'][:N_PROMPTS])
    np.save(base / 'lengths.npy', e['lengths'][:N_PROMPTS])
    np.save(base / 'reference.npy', e['reference'][:N_PROMPTS])
    np.save(base / 'hf_tokens.npy', e['hf_tokens'][:N_PROMPTS])
    return base, out


def build(base, ft, name, flags):
    eng = base / f'eng_{name}'
    if not (eng / 'config.json').exists():
        subprocess.run([sys.executable, os.path.join(EX, 'build.py'), '--model_dir', ft, '--output_dir', str(eng),
                        '--max_batch_size', str(BATCH), '--max_input_len', '256', '--max_output_len', str(NEW), '--log_level', 'error']
                       + flags, check=True, cwd=EX, timeout=900)
    return eng


def summarize(base, eng, out):
    r = subprocess.run([sys.executable, os.path.join(EX, 'summarize.py'), '--hf_model_location', FIX, '--test_hf', '--test_trt_llm',
                        '--hf_tokens_npy', str(base / 'hf_tokens.npy'), '--data_type', 'fp32', '--engine_dir', str(eng),
                        '--prompts_npy', str(base / 'prompts.npy'), '--prompt_lengths_npy', str(base / 'lengths.npy'),
                        '--references_npy', str(base / 'reference.npy'), '--output_len', str(NEW), '--batch_size', str(BATCH),
                        '--max_ite', str(N_PROMPTS // BATCH), '--log_level', 'error', '--output_json', str(out)],
                       cwd=EX, timeout=1800, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.load(open(out))


MISCALIBRATION = [1.25, 1.5, 2.0, 3.0]


def miscalibrated_ft_dir(base, good, factor):
    """the SmoothQuant static FT directory calibrated as if every activation range were 1 / factor of what it is"""
    bad = base / f'ft_bad_{factor}'
    shutil.copytree(good, bad)
    nx = 0
    for f in sorted(bad.glob('*scale_x_orig_quant.bin')):
        (np.fromfile(f, np.float32) * factor).astype(np.float32).tofile(f)
        nx += 1
    for f in sorted(list(bad.glob('*scale_y_accum_quant.bin')) + list(bad.glob('*scale_y_accum_quant.col.bin'))):
        (np.fromfile(f, np.float32) / factor).astype(np.float32).tofile(f)
    assert nx > 0
    return str(bad)


@pytest.fixture(scope='module')
def rouge_runs(ft_dirs):
    """build.py -> summarize.py for every configuration and every graded miscalibration, as CONCURRENT command-line pipelines (each is
    a process of its own with batch-8 engines - none of them takes the batch-1 one-launch decode path, which wants the chip to
    itself): {name or ('bad', factor): result dict or the exception}.  Run one after the other they were 110 s of the GPU suite."""
    base, ft = ft_dirs
    jobs = {name: (ft[CONFIGS[name][0]], CONFIGS[name][1]) for name in CONFIGS}
    for factor in MISCALIBRATION:
        jobs[('bad', factor)] = (miscalibrated_ft_dir(base, ft[True], factor), CONFIGS['sq_static_int8kv'][1])

    def run(item):
        key, (ftd, flags) = item
        tag = key if isinstance(key, str) else f'bad_{key[1]}'
        try:
            return key, summarize(base, build(base, ftd, tag, flags), base / f'rouge_{tag}.json')
        except BaseException as e:  # reported by the test that asks for this key
            return key, e

    with concurrent.futures.ThreadPoolExecutor(len(jobs)) as ex:
        return dict(ex.map(run, jobs.items()))


def result_of(rouge_runs, key):
    res = rouge_runs[key]
    if isinstance(res, BaseException):
        raise res
    return res


def verdict(res):
    lo, hi = res['rougeL_delta_ci95']
    return hi >= -1.0 and lo <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(CONFIGS))
def test_rouge_l_delta_vs_hf_within_one(rouge_runs, name):
    res = result_of(rouge_runs, name)
    lo, hi = res['rougeL_delta_ci95']
    print(f'[stochastic parent, {name}] ROUGE-L engine {res["tensorrt_llm"]["rougeL"]:.2f} / HF {res["hf"]["rougeL"]:.2f}: delta '
          f'{res["rougeL_delta_vs_hf"]:+.2f}, 95 % interval [{lo:+.2f}, {hi:+.2f}] over {res["samples"]} prompts; continuations identical '
          f'to HF\'s: {res["samples_identical_to_hf"]}/{res["samples"]}, token match {res["token_match_rate"]:.3f}; point estimate '
          f'{"inside" if abs(res["rougeL_delta_vs_hf"]) <= 1.0 else "OUTSIDE"} +-1')
    assert 30.0 <= res['hf']['rougeL'] <= 70.0  # the instrument is in its sensitive range
    assert verdict(res), res['rougeL_delta_ci95']
    if name != 'woq4_int8kv':
        # BASELINE.json's configurations (fp16, int8 KV, weight-only int8, SmoothQuant): the point estimate itself is inside +-1.
        # int4 weight-only is not one of them and is the one configuration whose point estimate is not (r05: -1.95, interval
        # [-4.20, +0.26], token match 0.47): the engine's teacher-forced error equals the numpy restatement's to three digits (2.0659
        # vs 2.0678, the test below), i.e. what is lost is lost by 4-bit weights on a 3.7 M-parameter model, not by the kernels
        assert abs(res['rougeL_delta_vs_hf']) <= 1.0, res['rougeL_delta_vs_hf']
    if name == 'fp16':  # an fp16 engine differs from HF fp32 on exact ties only
        assert res['token_match_rate'] > 0.9


@pytest.mark.gpu
@pytest.mark.parametrize('factor', MISCALIBRATION)
def test_the_criterion_fails_for_a_mildly_miscalibrated_engine(rouge_runs, factor):
    """Negative control, mild on purpose (r04's was a 24 x wrong KV scale): the SmoothQuant static engine calibrated as if every
    activation range were a THIRD of what it is - scale_x_orig_quant x 3 and, consistently, scale_y_accum_quant / 3, so values inside
    the (too small) range still dequantise correctly and only what exceeds it saturates at +-127.  The interval of the delta must lie
    entirely below -1.  (An int8 KV-cache scale 3 x too small does NOT fail - delta +0.15 [+0.02, +0.30], token match 0.96, measured
    r05 - and should not: less than a percent of the cached values lie above a third of their calibrated maximum.)"""
    res = result_of(rouge_runs, ('bad', factor))
    lo, hi = res['rougeL_delta_ci95']
    print(f'[stochastic parent, SmoothQuant static with activation ranges / {factor}] ROUGE-L engine {res["tensorrt_llm"]["rougeL"]:.2f} / HF '
          f'{res["hf"]["rougeL"]:.2f}: delta {res["rougeL_delta_vs_hf"]:+.2f}, 95 % interval [{lo:+.2f}, {hi:+.2f}], token match '
          f'{res["token_match_rate"]:.3f}')
    if factor >= 2.0:
        assert not verdict(res) and hi < -1.0, res['rougeL_delta_ci95']


@pytest.mark.gpu
@pytest.mark.parametrize('parent', ['deterministic', 'stochastic'])
@pytest.mark.parametrize('name', list(TP.ORACLE_MODES))
def test_teacher_forced_logits_within_k_times_the_algorithms_own_error(parent, name):
    """trained_parents.py: on HF's own token path, the engine's distance to HF fp32 is at most K = 1.25 x the distance of the numpy
    restatement of the same algorithm (same integers, same scales, the reference's rounding points) + the reference's fp16 allowance
    1e-1 - for every configuration, on both parents.  fp16 itself: the reference's atol 1e-1 (test_llama.py:288, 354)."""
    STEPS, NP = 48, 8
    cfg, qmodel = TP.quantised(parent, name)
    ids, lens, toks, hf_logits, scale = TP.teacher_forced_prompts(parent, NP)
    want = hf_logits[:, :STEPS]
    eng = TP.engine_logits(qmodel, cfg, ids, lens, toks, STEPS)
    alg = TP.oracle_logits(qmodel, ids, lens, toks, STEPS)
    e_err, a_err, ea = np.abs(eng - want), np.abs(alg - want), np.abs(eng - alg)
    print(f'[{parent} parent, {name}] teacher-forced over {len(lens)} prompts x {STEPS} steps (logit scale {scale:.1f}): engine vs HF max '
          f'{e_err.max():.4f} mean {e_err.mean():.5f}; algorithm (numpy restatement) vs HF max {a_err.max():.4f} mean {a_err.mean():.5f}; '
          f'engine vs algorithm max {ea.max():.4f} mean {ea.mean():.5f}')
    if name == 'fp16':
        assert e_err.max() <= TP.A_FP16
        return
    assert e_err.max() <= TP.K_ALGORITHM * a_err.max() + TP.A_FP16
    assert e_err.mean() <= TP.K_ALGORITHM * a_err.mean() + TP.A_FP16 / 10
    # arg-max agreement wherever HF's margin exceeds twice the engine's error at that step
    top2 = np.sort(want, axis=-1)[..., -2:]
    conf = (top2[..., 1] - top2[..., 0]) > 2 * e_err.max(-1)
    assert np.all((eng.argmax(-1) == want.argmax(-1))[conf])
