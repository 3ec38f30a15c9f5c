"""TEST INFRASTRUCTURE - CPU restatement of the reference's beam-search step and back-tracking (numpy).

Only tests/ may import this file; the product path never does.

What it restates (T = /root/reference/tensorrt_llm_july-release-v1):
  * candidate scores and selection: T/cpp/tensorrt_llm/kernels/onlineSoftmaxBeamsearchKernels.cu - per hypothesis a softmax
    over the vocabulary (:333-400; a FINISHED hypothesis gets probability 1 on end_id and 0 elsewhere, :349-366), score =
    cum_log_prob + log prob, then the beam_width best of the beam_width x vocab candidates of every batch entry
    (batch_topk_kernel, :116-330, the beam_hyps == nullptr branch the Python runtime takes: generation.py:949-961 passes no
    beam_hyps);
  * state update: T/cpp/tensorrt_llm/layers/onlineBeamSearchLayer.cu:28-61 (sequence length, finished) and
    T/cpp/tensorrt_llm/layers/baseBeamSearchLayer.cu update_indir_cache_kernel (the new hypothesis inherits its parent's
    cache indirection and points the newest slot at the parent);
  * gather_tree: T/cpp/tensorrt_llm/kernels/decodingKernels.cu:30-171.
Parity pinned by: tests/test_beam_oracle.py (hand-worked vectors + a brute-force exhaustive search on a toy model).
"""
import numpy as np


def log_softmax(x):
    x = x.astype(np.float64)
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def candidate_scores(cum, logits, finished, end_id):
    """cum [W], logits [W, V], finished [W] -> scores [W, V] (float64); -inf where a finished hypothesis cannot go."""
    W, V = logits.shape
    sc = cum.astype(np.float64)[:, None] + log_softmax(logits)
    for k in range(W):
        if finished[k]:
            sc[k, :] = -np.inf
            if end_id >= 0:
                sc[k, end_id] = cum[k]
    return sc


def beam_step(cum, logits, finished, end_id):
    """One step for one batch entry.  Returns (tokens [W], parents [W], new_cum [W], new_finished [W]), best first; ties go
    to the lowest flat index parent * V + token."""
    W, V = logits.shape
    sc = candidate_scores(cum, logits, finished, end_id).reshape(-1)
    order = np.lexsort((np.arange(sc.size), -sc))[:W]
    parents, tokens = order // V, order % V
    new_fin = np.array([bool(finished[p]) or (end_id >= 0 and t == end_id) for p, t in zip(parents, tokens)])
    return tokens.astype(np.int32), parents.astype(np.int32), sc[order], new_fin


def update_cache_indirection(ci, parents, slot, used):
    """ci [W, Smax]: new hypothesis j reads slots [0, used) where its parent did, and slot `slot` (the K/V of the token the
    parent consumed this step; None for the step that follows the prompt) from the parent's own rows."""
    new = ci.copy()
    for j, p in enumerate(parents):
        new[j, :used] = ci[p, :used]
        if slot is not None:
            new[j, slot] = p
    return new


def gather_tree(step_ids, parent_ids, last, first, end_id):
    """step_ids / parent_ids [W, Smax]; slots [first, last] hold generated tokens.  Returns [W, Smax] back-tracked rows;
    everything after the first end_id and beyond `last` is end_id (decodingKernels.cu:88-156)."""
    W, smax = step_ids.shape
    out = np.empty_like(step_ids)
    fill = end_id if end_id >= 0 else 0
    for j0 in range(W):
        out[j0, :first] = step_ids[j0, :first]
        j = j0
        for t in range(last, first - 1, -1):
            out[j0, t] = step_ids[j, t]
            j = parent_ids[j, t]
        out[j0, last + 1:] = fill
        if end_id >= 0:
            hit = np.where(out[j0, first:last + 1] == end_id)[0]
            if len(hit):
                out[j0, first + hit[0] + 1:] = end_id
    return out
