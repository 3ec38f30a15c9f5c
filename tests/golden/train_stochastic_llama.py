"""Trains the SECOND small LLaMA of the accuracy instrument - the one on which "ROUGE-L delta vs HF <= 1" can go either way -
and writes the fixture tests/golden/trained_llama_stochastic/.  Run in the BUILD container (CPU, ~25 min on 8 cores).

    python tests/golden/train_stochastic_llama.py [--steps 3500] [--out tests/golden/trained_llama_stochastic]

Why a second parent ("What's missing" 1): the reference decides "ROUGE within ~1" where HF itself scores ROUGE-L 15.2
against the highlights and an fp16 engine lands 1.5 away (T/README.md:912-921, T/examples/llama_quant/summarize.py:321-323) - a regime
full of near-ties.  The first trained parent (train_tiny_llama.py) speaks a DETERMINISTIC language: HF's top-1 / top-2 margin is >= 5.5
at every step, HF's ROUGE-L against the highlights is 100.0, and no quantisation error below ~2.8 logits can move a token - the
instrument is saturated.  The random 7B parent is the other extreme (margin < 0.1 on 57 % of the steps).  This parent sits between:

The language ("records and phrases", train_tiny_llama.Language) made STOCHASTIC at two kinds of position:
  * the successor of a phrase: for ~half of the 120 phrases the next phrase is one of 2 - 4 near-equiprobable alternatives
    (probabilities like 0.38 / 0.33 / 0.29; half of those alternatives skip one or two phrases ahead on the same cycle - a summary
    that drops a sentence - the other half jump elsewhere), for the rest it is perm[p] with probability 0.97;
  * the body of a phrase: ~40 % of the body positions are a choice between TWO pool tokens (0.55 / 0.45), independent of everything
    else - the "synonyms": flipping one changes one token and nothing after it.
  Look-ups (a key token followed by the value the document's record binds to it) stay deterministic: they are what the model needs
  its attention and its KV cache for, and what a broken cache gets wrong.
So >= 25 % of the positions have 2 - 4 continuations within log(0.55 / 0.45) = 0.2 logits of each other; the most likely
continuation (the arg-max path HF's greedy search follows) and a SAMPLED continuation (the `highlights`: one draw from the
language, as a human summary is one draw from what people write) agree only in part - HF's own ROUGE-L is 30 - 70, not 100.

Evaluation set: 256 prompts x 100 new tokens (the reference: 20 x 100), so that the per-prompt noise of ROUGE-L after a flipped
near-tie averages out and the delta can be quoted with a bootstrap interval.

Fixture (all seeded; regenerate with this script):
  config.json, model.safetensors   HF LlamaForCausalLM, fp16 weights (same architecture as trained_llama)
  eval.npz   prompts [256, Lmax] + lengths, reference (sampled) continuations [256, 100], the language's arg-max continuations
             [256, 100], HF fp32 greedy continuations [256, 100], HF fp32 logits on HF's own path for the first 8 prompts
             [8, 100, 512], calibration prompts [64, 192]
  TRAINLOG.json   loss curve, HF margins (fraction below 1.0 / 0.2), HF ROUGE-L against the references, fraction of stochastic
             positions of the language
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import train_tiny_llama as T  # noqa: E402  (the deterministic language, the model configuration and the training loop)

NEW = 100


class StochasticLanguage(T.Language):
    def __init__(self, seed=4321):
        super().__init__(seed=1234)  # the same phrases, keys and permutation as the deterministic parent
        r = np.random.default_rng(seed)
        # ---- successors
        self.succ, self.succ_p = [], []
        tables = {2: [0.55, 0.45], 3: [0.38, 0.33, 0.29], 4: [0.30, 0.26, 0.23, 0.21]}
        for p in range(T.NPHRASE):
            if r.random() < 0.5:
                k = int(r.integers(2, 5))
                if r.random() < 0.5:  # skip ahead on the same cycle
                    alts, q = [], p
                    for _ in range(k):
                        q = int(self.perm[q])
                        alts.append(q)
                else:
                    alts = [int(self.perm[p])]
                    while len(alts) < k:
                        c = int(r.integers(0, T.NPHRASE))
                        if c not in alts and c != p:
                            alts.append(c)
                order = r.permutation(k)  # which alternative is the most likely one is random
                self.succ.append([alts[i] for i in order])
                self.succ_p.append(tables[k])
            else:
                self.succ.append([int(self.perm[p])])
                self.succ_p.append([1.0])
        # ---- body alternatives: position j of phrase p is body[p][j] (0.55) or alt[p][j] (0.45); -1 = no alternative
        self.alt = []
        for p in range(T.NPHRASE):
            a = []
            for tok in self.body[p]:
                if r.random() < 0.4:
                    c = int(r.integers(T.POOL0, T.POOL0 + T.NPOOL))
                    while c == tok:
                        c = int(r.integers(T.POOL0, T.POOL0 + T.NPOOL))
                    a.append(c)
                else:
                    a.append(-1)
            self.alt.append(a)

    # r = None -> the arg-max choice everywhere
    def phrase(self, p, binding, r=None):
        t = [T.HEAD0 + p]
        for tok, alt in zip(self.body[p], self.alt[p]):
            t.append(alt if (alt >= 0 and r is not None and r.random() < 0.45) else tok)
        if self.key[p] >= 0:
            t += [T.KEY0 + self.key[p], T.VAL0 + binding[self.key[p]]]
        return t

    def next_phrase(self, p, r=None):
        if r is None:
            return self.succ[p][0]
        if len(self.succ[p]) == 1:
            return self.succ[p][0] if r.random() < 0.97 else int(r.integers(0, T.NPHRASE))
        return self.succ[p][int(r.choice(len(self.succ[p]), p=self.succ_p[p]))]

    def document(self, r, length=T.DOC_LEN, n_queries=None):
        binding = r.integers(0, T.NVAL, T.NKEY)
        order = r.permutation(T.NKEY)
        t = [T.BOS]
        for k in order:
            t.append(T.REC0 + int(k) * T.NVAL + int(binding[k]))
        t.append(T.SEP)
        nq = int(r.integers(0, 49)) if n_queries is None else n_queries
        for k in r.integers(0, T.NKEY, nq):
            t += [T.KEY0 + int(k), T.VAL0 + int(binding[k])]
        t.append(T.SEP2)
        self.body0 = len(t)
        p = int(r.integers(0, T.NPHRASE))
        starts = []
        while len(t) < length:
            starts.append((len(t), p))
            t += self.phrase(p, binding, r)
            p = self.next_phrase(p, r)
        return np.array(t[:length], np.int64), binding, starts

    def continuation(self, doc, binding, starts, cut, n, r=None):
        """Continuation of doc[:cut]: the rest of the open phrase (its arg-max tokens when r is None, else as the document has
        it), then phrases chosen by arg-max (r None) or sampled (r a generator)."""
        pos, p = [(s, q) for s, q in starts if s < cut][-1]
        plen = len(self.phrase(p, binding))
        if r is None:
            out = self.phrase(p, binding)[cut - pos:]
        else:
            out = doc[cut:pos + plen].tolist()
        while len(out) < n:
            p = self.next_phrase(p, r)
            out += self.phrase(p, binding, r)
        return np.array(out[:n], np.int64)

    def stochastic_fraction(self):
        """Fraction of body-region positions with >= 2 near-equiprobable continuations (phrase-length weighted, uniform phrases)."""
        tot, st = 0, 0
        for p in range(T.NPHRASE):
            n = 1 + len(self.body[p]) + (2 if self.key[p] >= 0 else 0)
            tot += n
            st += sum(1 for a in self.alt[p] if a >= 0)
        # the head of the NEXT phrase is the stochastic position of a phrase with several successors
        st += sum(1 for p in range(T.NPHRASE) if len(self.succ[p]) > 1)
        return st / tot


def rouge_l_ids(a, b):
    sys.path.insert(0, os.path.join(HERE, '..', '..', 'trtllm-llama_amd', 'examples', 'llama_quant'))
    from summarize import _f, _lcs
    a, b = [int(x) for x in a], [int(x) for x in b]
    return _f(_lcs(a, b), len(a), len(b))


def eval_set(lang, n_eval, seed=99):
    re = np.random.default_rng(seed)
    rs = np.random.default_rng(seed + 1)
    prompts, refs, argmax, lens = [], [], [], []
    for _ in range(n_eval):
        doc, binding, starts = lang.document(re, length=480, n_queries=int(re.integers(4, 25)))
        # the cut sits on a phrase boundary + 1 (right behind a head token), so the prompt fixes which phrase is open
        cands = [s for s, _ in starts if lang.body0 + 20 <= s < lang.body0 + 90]
        cut = int(cands[int(re.integers(0, len(cands)))]) + 1
        prompts.append(doc[:cut])
        lens.append(cut)
        refs.append(lang.continuation(doc, binding, starts, cut, NEW, rs))
        argmax.append(lang.continuation(doc, binding, starts, cut, NEW, None))
    return prompts, lens, np.stack(refs), np.stack(argmax)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=3500)
    ap.add_argument('--batch', type=int, default=48)
    ap.add_argument('--lr', type=float, default=3e-3)
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--out', default=os.path.join(HERE, 'trained_llama_stochastic'))
    ap.add_argument('--n_eval', type=int, default=256)
    ap.add_argument('--n_logits', type=int, default=8)
    ap.add_argument('--eval_only', action='store_true')
    ap.add_argument('--language_only', action='store_true', help='print the language-level statistics and exit (no model)')
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    torch.manual_seed(0)
    lang = StochasticLanguage()
    prompts, lens, refs, argmax = eval_set(lang, args.n_eval)
    lang_rouge = float(np.mean([rouge_l_ids(a, b) for a, b in zip(argmax, refs)])) * 100
    print(f'language: stochastic positions {lang.stochastic_fraction():.3f}; ROUGE-L of the arg-max continuation against the sampled one '
          f'{lang_rouge:.1f}; prompts {min(lens)} .. {max(lens)} tokens', flush=True)
    if args.language_only:
        return
    from transformers import LlamaForCausalLM
    log = []
    t0 = time.time()
    if args.eval_only:
        model = LlamaForCausalLM.from_pretrained(args.out).float().eval()
        prev = json.load(open(os.path.join(args.out, 'TRAINLOG.json')))
        log, t0 = prev.get('loss', []), time.time() - prev.get('seconds', 0.0)
    else:
        model = T.train(args, lang, log, t0)
    lmax = max(lens)
    P = np.full((args.n_eval, lmax), T.PAD, np.int32)
    for i, p in enumerate(prompts):
        P[i, :len(p)] = p
    hf_out = np.zeros((args.n_eval, NEW), np.int32)
    margins = np.zeros((args.n_eval, NEW), np.float32)
    hf_logits = np.zeros((args.n_logits, NEW, T.V), np.float32)
    absmax = 0.0
    with torch.no_grad():
        for i, p in enumerate(prompts):
            ids = torch.from_numpy(p)[None]
            o = model(input_ids=ids, use_cache=True)
            past = o.past_key_values
            lg = o.logits[0, -1]
            for s in range(NEW):
                if i < args.n_logits:
                    hf_logits[i, s] = lg.numpy()
                absmax = max(absmax, float(lg.abs().max()))
                top = torch.topk(lg, 2).values
                margins[i, s] = float(top[0] - top[1])
                nxt = int(lg.argmax())
                hf_out[i, s] = nxt
                o = model(input_ids=torch.tensor([[nxt]]), past_key_values=past, use_cache=True)
                past = o.past_key_values
                lg = o.logits[0, -1]
    hf_rouge = [rouge_l_ids(a, b) * 100 for a, b in zip(hf_out, refs)]
    re = np.random.default_rng(5)
    calib = np.stack([lang.document(re, length=192)[0] for _ in range(64)]).astype(np.int32)
    np.savez_compressed(os.path.join(args.out, 'eval.npz'), prompts=P, lengths=np.array(lens, np.int32), reference=refs.astype(np.int32),
                        language_argmax=argmax.astype(np.int32), hf_tokens=hf_out, hf_logits=hf_logits, hf_logits_absmax=np.float32(absmax),
                        hf_margins=margins, calib=calib)
    info = dict(steps=len(log) and log[-1][0] + 1 or args.steps, batch=args.batch, doc_len=T.DOC_LEN, seconds=time.time() - t0, loss=log,
                n_eval=args.n_eval, new_tokens=NEW,
                language=dict(stochastic_position_fraction=lang.stochastic_fraction(), rougeL_argmax_vs_sampled=lang_rouge),
                hf_rougeL_vs_reference=dict(mean=float(np.mean(hf_rouge)), std_per_prompt=float(np.std(hf_rouge))),
                hf_greedy_vs_language_argmax_token_accuracy=float(np.mean(hf_out == argmax)),
                margin=dict(median=float(np.median(margins)), frac_below_1p0=float(np.mean(margins < 1.0)),
                            frac_below_0p2=float(np.mean(margins < 0.2)), min=float(margins.min())),
                logit_absmax=absmax,
                distinct_tokens_per_continuation=float(np.mean([len(set(x.tolist())) for x in hf_out])),
                versions=dict(torch=torch.__version__, transformers=__import__('transformers').__version__))
    with open(os.path.join(args.out, 'TRAINLOG.json'), 'w') as f:
        json.dump(info, f, indent=1)
    print(json.dumps({k: v for k, v in info.items() if k != 'loss'}, indent=1))


if __name__ == '__main__':
    main()
