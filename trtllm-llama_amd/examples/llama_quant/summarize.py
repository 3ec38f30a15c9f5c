"""Accuracy check of an engine against HF: summarisation + ROUGE (T/examples/llama_quant/summarize.py).

Same command line as the reference.  With a HF tokenizer and an offline copy of ccdv/cnn_dailymail (--dataset_path) it
does what the reference does: first `max_ite` test articles, prompt = article + ' TL;DR: ' truncated to 923 tokens, 100
new tokens, top-k = 1, ROUGE of the engine's and of HF's summaries against the highlights.  ROUGE is computed in this
file (rouge_score / datasets.load_metric are not available offline): rouge1 / rouge2 / rougeL / rougeLsum F-measures with
rouge_score's default tokenisation (lower-case, runs of [a-z0-9], no stemming), averaged over the samples.

Without dataset or tokenizer (--prompts_npy, or nothing at all -> seeded random token prompts) the texts are the token-id
strings themselves: ROUGE of the engine's continuation against HF's continuation of the same prompt, plus the token match
rate - the "ROUGE-L delta vs HF" of BASELINE.json measured without a dataset.
"""
import argparse
import copy
import json
import os
import re

import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))

import tensorrt_llm  # noqa: E402
import tensorrt_llm.profiler as profiler  # noqa: E402
from tensorrt_llm.logger import logger  # noqa: E402

from run import load_session  # noqa: E402  (same directory)


# ---------------------------------------------------------------------------------------------------------- ROUGE
def _tokens(text):
    return re.findall(r'[a-z0-9]+', text.lower())


def _ngrams(toks, n):
    c = {}
    for i in range(len(toks) - n + 1):
        g = tuple(toks[i:i + n])
        c[g] = c.get(g, 0) + 1
    return c


def _f(match, n_pred, n_ref):
    if n_pred == 0 or n_ref == 0 or match == 0:
        return 0.0
    p, r = match / n_pred, match / n_ref
    return 2 * p * r / (p + r)


def rouge_n(pred, ref, n):
    a, b = _ngrams(_tokens(pred), n), _ngrams(_tokens(ref), n)
    match = sum(min(v, b.get(g, 0)) for g, v in a.items())
    return _f(match, sum(a.values()), sum(b.values()))


def _lcs(a, b):
    if not a or not b:
        return 0
    prev = [0] * (len(b) + 1)
    for x in a:
        cur = [0]
        for j, y in enumerate(b):
            cur.append(prev[j] + 1 if x == y else max(prev[j + 1], cur[j]))
        prev = cur
    return prev[-1]


def rouge_l(pred, ref):
    a, b = _tokens(pred), _tokens(ref)
    return _f(_lcs(a, b), len(a), len(b))


def _lcs_table(a, b):
    t = [[0] * (len(b) + 1) for _ in range(len(a) + 1)]
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            t[i + 1][j + 1] = t[i][j] + 1 if x == y else max(t[i][j + 1], t[i + 1][j])
    return t


def rouge_lsum(pred, ref):
    """Summary-level LCS (union LCS over the reference's lines, rouge_score's rougeLsum)."""
    ps = [_tokens(s) for s in pred.split('\n') if _tokens(s)]
    rs = [_tokens(s) for s in ref.split('\n') if _tokens(s)]
    n_pred, n_ref = sum(map(len, ps)), sum(map(len, rs))
    if not n_pred or not n_ref:
        return 0.0
    pc, rc = {}, {}
    for s in ps:
        for w in s:
            pc[w] = pc.get(w, 0) + 1
    for s in rs:
        for w in s:
            rc[w] = rc.get(w, 0) + 1
    hits = 0
    for r in rs:
        union = set()
        for p_ in ps:
            t = _lcs_table(r, p_)
            i, j = len(r), len(p_)
            while i > 0 and j > 0:
                if r[i - 1] == p_[j - 1]:
                    union.add(i - 1)
                    i, j = i - 1, j - 1
                elif t[i - 1][j] >= t[i][j - 1]:
                    i -= 1
                else:
                    j -= 1
        for i in sorted(union):
            w = r[i]
            if pc.get(w, 0) > 0 and rc.get(w, 0) > 0:
                hits += 1
                pc[w] -= 1
                rc[w] -= 1
    return _f(hits, n_pred, n_ref)


class Rouge:
    """Accumulates (prediction, reference) pairs; compute() = mean F-measure x 100 per ROUGE type."""
    KEYS = ('rouge1', 'rouge2', 'rougeL', 'rougeLsum')

    def __init__(self):
        self.rows = []

    def add_batch(self, predictions, references):
        for p, r in zip(predictions, references):
            self.rows.append((rouge_n(p, r, 1), rouge_n(p, r, 2), rouge_l(p, r), rouge_lsum(p, r)))

    def compute(self):
        if not self.rows:
            return {k: 0.0 for k in self.KEYS}
        m = np.mean(np.array(self.rows), axis=0) * 100
        return dict(zip(self.KEYS, m.tolist()))


# ---------------------------------------------------------------------------------------------------- main flow
def ids_to_text(ids):
    return ' '.join(f't{int(i)}' for i in ids)


def main(args):
    runtime_rank = tensorrt_llm.mpi_rank()
    logger.set_level(args.log_level)
    test_hf = args.test_hf and runtime_rank == 0  # only run hf on rank 0
    test_trt_llm = args.test_trt_llm
    output_len, test_token_num = args.output_len, args.max_input_tokens
    import torch

    tokenizer, dataset = None, None
    if args.hf_model_location and not args.prompts_npy and not args.synthetic:
        try:
            from transformers import LlamaTokenizer
            tokenizer = LlamaTokenizer.from_pretrained(args.hf_model_location, legacy=False, padding_side='left')
            tokenizer.pad_token = tokenizer.eos_token
            from datasets import load_dataset
            dataset = load_dataset('ccdv/cnn_dailymail', '3.0.0', cache_dir=args.dataset_path)['test']
        except Exception as e:
            logger.warning(f'tokenizer / cnn_dailymail unavailable ({e!r}): falling back to token-id prompts')
            tokenizer, dataset = None, None
    pad_id = end_id = 2
    if tokenizer is not None:
        pad_id = tokenizer.encode(tokenizer.pad_token, add_special_tokens=False)[0]
        end_id = tokenizer.encode(tokenizer.eos_token, add_special_tokens=False)[0]

    decoder = None
    vocab = None
    if test_trt_llm:
        decoder, _ = load_session(args.engine_dir)
        with open(os.path.join(args.engine_dir, 'config.json')) as f:
            vocab = json.load(f)['builder_config']['vocab_size']
    model = None
    hf_precomputed = None
    if test_hf and args.hf_tokens_npy:
        # HF's greedy continuations of the same prompts, made once with the fixture (tests/golden/train_stochastic_llama.py: HF fp32
        # on the CPU, the reference's run_hf.py path) - several engines are then scored against ONE HF run instead of repeating it
        hf_precomputed = np.load(args.hf_tokens_npy)
    if test_hf and hf_precomputed is None:
        from transformers import AutoModelForCausalLM
        profiler.start('load HF model')
        model = AutoModelForCausalLM.from_pretrained(args.hf_model_location)
        profiler.stop('load HF model')
        # explicit either way: newer transformers load a checkpoint in the dtype it was saved in
        model = model.half() if args.data_type == 'fp16' else model.float()
        model.to('cuda' if torch.cuda.is_available() else 'cpu').eval()
        vocab = vocab or model.config.vocab_size

    # ---- prompts: list of int32 arrays (token ids) + reference texts (or None)
    prompts, references = [], []
    if dataset is not None:
        for i in range(min(len(dataset), args.max_ite * args.batch_size)):
            line = (dataset[i]['article'] + ' TL;DR: ').strip().replace(" n't", "n't")
            ids = tokenizer.encode(line, add_special_tokens=False)[:test_token_num]
            prompts.append(np.array(ids, np.int32))
            references.append(dataset[i]['highlights'])
    elif args.prompts_npy:
        arr = np.load(args.prompts_npy)
        n = args.max_ite * args.batch_size
        if args.prompt_lengths_npy:  # ragged prompts in a padded [n, Lmax] array
            plens = np.load(args.prompt_lengths_npy)
            prompts = [np.asarray(r, np.int32)[:int(l)][:test_token_num] for r, l in zip(arr, plens)][:n]
        else:
            prompts = [np.asarray(r, np.int32)[:test_token_num] for r in arr][:n]
        references = [None] * len(prompts)
        if args.references_npy:  # the dataset's `highlights`, as token ids [n, L]: the texts are the token-id strings
            references = [ids_to_text(r) for r in np.load(args.references_npy)][:n]
    else:
        rng = np.random.default_rng(1)
        n = args.max_ite * args.batch_size
        lens = rng.integers(max(8, args.synthetic_len // 2), args.synthetic_len + 1, n)
        prompts = [rng.integers(3, vocab, int(l)).astype(np.int32) for l in lens]
        references = [None] * n

    def summarize_tensorrt_llm(batch):
        lens = np.array([len(p) for p in batch], np.int32)
        max_len = int(lens.max())
        ids = np.full((len(batch), max_len), pad_id, np.int32)
        for i, p in enumerate(batch):
            ids[i, :len(p)] = p  # right padding + input_lengths, as GenerationSession expects
        decoder.setup(len(batch), max_len, output_len)
        from tensorrt_llm.runtime import SamplingConfig
        out = np.asarray(decoder.decode(ids, lens, SamplingConfig(end_id=end_id if args.stop_at_eos else -1, pad_id=pad_id,
                                                                  num_beams=args.num_beams, top_k=args.top_k)))
        # every sequence continues right after its own last prompt token slot max_len (padded layout)
        return [out[i, 0, max_len:max_len + output_len] for i in range(len(batch))]

    @torch.no_grad()
    def summarize_hf(batch):
        outs = []
        for p in batch:  # one by one: no padding ambiguity
            ids = torch.from_numpy(p.astype(np.int64))[None].to(model.device)
            o = model.generate(ids, max_new_tokens=output_len, do_sample=False, num_beams=args.num_beams,
                               eos_token_id=end_id if args.stop_at_eos else None, pad_token_id=pad_id)
            o = o[0, len(p):].cpu().numpy()
            outs.append(np.pad(o, (0, output_len - len(o)), constant_values=pad_id))
        return outs

    def to_text(tok_ids):
        if tokenizer is not None:
            return tokenizer.decode([int(t) for t in tok_ids], skip_special_tokens=True)
        return ids_to_text(tok_ids)

    metric_trt, metric_hf, metric_vs_hf = Rouge(), Rouge(), Rouge()
    match, total = 0, 0
    for it in range(0, len(prompts), args.batch_size):
        batch = prompts[it:it + args.batch_size]
        refs = references[it:it + args.batch_size]
        s_trt = s_hf = None
        if test_trt_llm:
            profiler.start('tensorrt_llm')
            s_trt = summarize_tensorrt_llm(batch)
            profiler.stop('tensorrt_llm')
        if test_hf:
            profiler.start('hf')
            s_hf = [hf_precomputed[it + i, :output_len] for i in range(len(batch))] if hf_precomputed is not None else summarize_hf(batch)
            profiler.stop('hf')
        if runtime_rank != 0:
            continue
        for i in range(len(batch)):
            if s_trt is not None and refs[i] is not None:
                metric_trt.add_batch([to_text(s_trt[i])], [refs[i]])
            if s_hf is not None and refs[i] is not None:
                metric_hf.add_batch([to_text(s_hf[i])], [refs[i]])
            if s_trt is not None and s_hf is not None:
                metric_vs_hf.add_batch([to_text(s_trt[i])], [to_text(s_hf[i])])
                match += int(np.sum(np.asarray(s_trt[i]) == np.asarray(s_hf[i])))
                total += len(s_hf[i])
                logger.debug(f'engine: {s_trt[i].tolist()}')
                logger.debug(f'hf    : {s_hf[i].tolist()}')

    result = {}
    if runtime_rank == 0:
        if test_trt_llm:
            logger.info(f'TensorRT-LLM (total latency: {profiler.elapsed_time_in_sec("tensorrt_llm")} sec)')
            if metric_trt.rows:
                result['tensorrt_llm'] = metric_trt.compute()
                logger.info('TensorRT-LLM beam 0 result')
                for k, v in result['tensorrt_llm'].items():
                    logger.info(f'  {k} : {v}')
        if test_hf:
            logger.info(f'Hugging Face (total latency: {profiler.elapsed_time_in_sec("hf")} sec)')
            if metric_hf.rows:
                result['hf'] = metric_hf.compute()
                logger.info('HF beam 0 result')
                for k, v in result['hf'].items():
                    logger.info(f'  {k} : {v}')
        if metric_vs_hf.rows:
            result['tensorrt_llm_vs_hf'] = metric_vs_hf.compute()
            result['token_match_rate'] = match / max(total, 1)
            logger.info('TensorRT-LLM summaries scored against the HF summaries')
            for k, v in result['tensorrt_llm_vs_hf'].items():
                logger.info(f'  {k} : {v}')
            logger.info(f'  token match rate : {result["token_match_rate"]:.4f}')
        if 'tensorrt_llm' in result and 'hf' in result:
            result['rougeL_delta_vs_hf'] = result['tensorrt_llm']['rougeL'] - result['hf']['rougeL']
            logger.info(f'  rougeL delta vs HF : {result["rougeL_delta_vs_hf"]:.3f}')
            # paired bootstrap over the prompts (seeded): how far the delta of THIS sample of prompts can be from the delta of the
            # prompt population.  The reference quotes a single number on 20 articles (README.md:921 "within about 1"); with
            # near-tied continuations one flipped token changes a whole summary, so the interval is what makes the number readable
            a = np.array([r[2] for r in metric_trt.rows]) * 100
            b = np.array([r[2] for r in metric_hf.rows]) * 100
            d = a - b
            rs = np.random.default_rng(args.bootstrap_seed)
            boots = d[rs.integers(0, len(d), (args.bootstrap, len(d)))].mean(1) if len(d) > 1 and args.bootstrap > 0 else np.array([d.mean()])
            result['rougeL_delta_ci95'] = [float(np.quantile(boots, 0.025)), float(np.quantile(boots, 0.975))]
            result['rougeL_delta_stderr'] = float(d.std(ddof=1) / np.sqrt(len(d))) if len(d) > 1 else 0.0
            result['samples'] = int(len(d))
            result['samples_identical_to_hf'] = int(np.sum([r[2] >= 1.0 - 1e-12 for r in metric_vs_hf.rows])) if metric_vs_hf.rows else 0
            if args.per_sample:
                result['per_sample_rougeL'] = dict(tensorrt_llm=a.tolist(), hf=b.tolist())
            logger.info(f'  rougeL delta 95 % interval (paired bootstrap over {len(d)} prompts): {result["rougeL_delta_ci95"]}')
        if args.output_json:  # before the checks: a run that fails them still leaves its numbers
            with open(args.output_json, 'w') as f:
                json.dump(result, f, indent=1)
        print(json.dumps(result))
        if args.check_accuracy:
            key = 'tensorrt_llm' if 'tensorrt_llm' in result else 'tensorrt_llm_vs_hf'
            assert result[key]['rouge1'] > args.tensorrt_llm_rouge1_threshold, result
            # the team's acceptance criterion, "ROUGE difference within about 1" (README.md:921), when HF ran beside the engine
            if args.rougeL_delta_threshold is not None and 'rougeL_delta_vs_hf' in result:
                assert abs(result['rougeL_delta_vs_hf']) <= args.rougeL_delta_threshold, \
                    {k: v for k, v in result.items() if k != 'per_sample_rougeL'}
    return result


def parse_arguments(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('--hf_model_location', type=str, default='/code/tensorrt_llm/models/llama-models/llama-7b-hf')
    parser.add_argument('--test_hf', action='store_true')
    parser.add_argument('--test_trt_llm', action='store_true')
    parser.add_argument('--data_type', type=str, choices=['fp32', 'fp16'], default='fp32')
    parser.add_argument('--dataset_path', type=str, default='')
    parser.add_argument('--log_level', type=str, default='info')
    parser.add_argument('--engine_dir', type=str, default='llama_outputs')
    parser.add_argument('--batch_size', type=int, default=1)
    parser.add_argument('--max_ite', type=int, default=20)
    parser.add_argument('--check_accuracy', action='store_true')
    parser.add_argument('--tensorrt_llm_rouge1_threshold', type=float, default=15.0)
    parser.add_argument('--num_beams', type=int, default=1)
    parser.add_argument('--top_k', type=int, default=1)
    # additions for boxes without the dataset / tokenizer
    parser.add_argument('--prompts_npy', type=str, default=None, help='int token-id prompts [n, L] instead of cnn_dailymail')
    parser.add_argument('--prompt_lengths_npy', type=str, default=None, help='int lengths [n] of the (padded) --prompts_npy rows')
    parser.add_argument('--references_npy', type=str, default=None,
                        help='reference continuations [n, L] (token ids) for --prompts_npy: ROUGE of the engine and of HF against them')
    parser.add_argument('--rougeL_delta_threshold', type=float, default=None,
                        help='with --check_accuracy: |rougeL(engine) - rougeL(HF)| against the references must not exceed this')
    parser.add_argument('--hf_tokens_npy', type=str, default=None,
                        help='with --test_hf: HF greedy continuations [n, >= output_len] made beforehand (same prompts, same order) '
                             'instead of running HF here')
    parser.add_argument('--bootstrap', type=int, default=2000, help='paired-bootstrap resamples for the ROUGE-L delta interval')
    parser.add_argument('--bootstrap_seed', type=int, default=0)
    parser.add_argument('--per_sample', action='store_true', help='keep the per-prompt ROUGE-L of both sides in --output_json')
    parser.add_argument('--synthetic', action='store_true', help='seeded random token prompts')
    parser.add_argument('--synthetic_len', type=int, default=64)
    parser.add_argument('--output_len', type=int, default=100)
    parser.add_argument('--max_input_tokens', type=int, default=923)
    parser.add_argument('--stop_at_eos', action='store_true')
    parser.add_argument('--output_json', type=str, default=None)
    return parser.parse_args(argv)


if __name__ == '__main__':
    main(parse_arguments())
