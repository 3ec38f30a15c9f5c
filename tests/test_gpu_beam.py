"""GPU parity of beam search (the sampler side of the generation loop, generation.py:823-997 with
num_beams > 1): the device beam step (candidate selection, cum_log_probs, finished handling, cache-indirection re-parenting),
the generation attention reading sibling hypotheses' cache rows through cache_indirection, and gather_tree.

Checked step by step against oracle/beam_oracle.py (itself pinned against Hugging Face's beam search on the CPU,
tests/test_beam_oracle.py): the oracle is fed the session's own logits, so every decision is comparable without
accumulating fp16 noise; the logits in turn are checked against the model oracle evaluated on each hypothesis'
back-tracked token sequence - which only holds if the attention followed the cache indirection correctly."""
import numpy as np
import pytest

from oracle import beam_oracle as BO
from oracle import quant_oracle as QO
from tensorrt_llm.runtime.native import NativeSession
from test_gpu_session import synth_model

pytestmark = pytest.mark.gpu


def make_session(cfg, w, mode, int8_kv, ids, lens):
    qmodel = QO.quantise_model(cfg, w, mode, int8_kv, calib_ids=ids, calib_lens=lens)
    s = NativeSession(dict(cfg, quant_mode=qmodel['quant_mode']))
    for k, v in qmodel['engine_tensors'].items():
        s.set_tensor(k, v)
    s.finalize()
    return s, qmodel


def prompts(cfg, B, S, lens, seed):
    r = np.random.default_rng(seed)
    ids = np.full((B, S), 2, np.int32)
    for b in range(B):
        ids[b, :lens[b]] = r.integers(3, cfg['vocab_size'], lens[b])
    return ids


class Tracker:
    """The oracle's beam state for every batch entry, advanced with the GPU's own choices after each comparison."""

    def __init__(self, Bc, W, S, smax, ids, end_id):
        self.Bc, self.W, self.S, self.smax, self.end_id = Bc, W, S, smax, end_id
        self.cum = np.full((Bc, W), -1e20)
        self.cum[:, 0] = 0.0
        self.fin = np.zeros((Bc, W), bool)
        self.ci = np.zeros((Bc, W, smax), np.int32)
        self.step_ids = np.zeros((Bc, W, smax), np.int32)
        self.step_ids[:, :, :S] = ids[:, None, :]
        self.parents = np.zeros((Bc, W, smax), np.int32)
        self.slot = S  # slot the next chosen token goes to

    def check(self, sess, logits, first):
        """logits [Bc, W, V] the step was taken on; compares the GPU's new state with the oracle's step and adopts it."""
        Bc, W = self.Bc, self.W
        raw = sess.output_ids().reshape(Bc, W, self.smax)
        st = sess.beam_state()
        par = st['parent_ids'].reshape(Bc, W, self.smax)[:, :, self.slot]
        tok = raw[:, :, self.slot]
        _, cum_gpu = sess.beam_output()
        assert (st['sequence_lengths'] == self.slot).all()
        for b in range(Bc):
            sc = BO.candidate_scores(self.cum[b], logits[b], self.fin[b], self.end_id)
            o_tok, o_par, o_cum, _ = BO.beam_step(self.cum[b], logits[b], self.fin[b], self.end_id)
            # scores of the GPU's picks == the oracle's best W scores, in order (robust to exact ties / fp32 rounding)
            picked = sc[par[b], tok[b]]
            np.testing.assert_allclose(picked, o_cum, atol=2e-3, rtol=1e-5)
            np.testing.assert_allclose(cum_gpu[b], picked, atol=2e-3, rtol=1e-5)
            assert len({(int(p), int(t)) for p, t in zip(par[b], tok[b])}) == W, 'duplicate candidates'
            clear = np.abs(np.diff(np.sort(sc.reshape(-1))[::-1][:W + 1])).min() > 5e-3
            if clear:  # no near-ties among the leaders: the picks themselves must agree
                np.testing.assert_array_equal(tok[b], o_tok)
                np.testing.assert_array_equal(par[b], o_par)
            # adopt the GPU's choice
            used = self.slot if first else self.slot - 1
            self.ci[b] = BO.update_cache_indirection(self.ci[b], par[b], None if first else self.slot - 1, used)
            self.fin[b] = np.array([self.fin[b][p] or (self.end_id >= 0 and t == self.end_id) for p, t in zip(par[b], tok[b])])
            self.cum[b] = picked
            self.step_ids[b, :, self.slot] = tok[b]
            self.parents[b, :, self.slot] = par[b]
            got_ci = st['cache_indirection'].reshape(Bc, W, self.smax)[b]
            n_valid = self.slot if not first else self.S
            np.testing.assert_array_equal(got_ci[:, :n_valid], self.ci[b][:, :n_valid])
            np.testing.assert_array_equal(st['finished'].reshape(Bc, W)[b].astype(bool), self.fin[b])
        self.slot += 1

    def sequences(self):
        return np.stack([BO.gather_tree(self.step_ids[b], self.parents[b], self.slot - 1, self.S, self.end_id)
                         for b in range(self.Bc)])


@pytest.mark.parametrize('mode,int8_kv,Bc,W', [('fp16', 0, 2, 3), ('fp16', 0, 1, 8), ('sq_static_pc', 1, 2, 4), ('woq8', 0, 1, 2)])
def test_beam_search_step_by_step(mode, int8_kv, Bc, W):
    cfg, w = synth_model(71)
    S, NEW = 10, 7
    lens = np.array([10, 6][:Bc], np.int32)
    ids = prompts(cfg, Bc, S, lens, 3)
    s, qmodel = make_session(cfg, w, mode, int8_kv, ids, lens)
    V, smax = cfg['vocab_size'], S + NEW
    s.setup(Bc, S, NEW, beam_width=W)
    s.context(ids, lens)
    tr = Tracker(Bc, W, S, smax, ids, end_id=-1)
    lg = s.logits()
    assert lg.shape == (Bc, V)
    ref0, _ = QO.run_model(qmodel, ids, lens, 1)
    scale = max(np.abs(ref0[0]).max(), 1.0)
    sq = mode.startswith('sq')
    np.testing.assert_allclose(lg, ref0[0], atol=(8e-2 if sq else 3e-2) * scale)
    tr.check(s, np.repeat(lg[:, None, :], W, axis=1), first=True)
    for step in range(NEW - 1):
        seqs = tr.sequences()  # hypotheses the next logits belong to
        s.step(1, use_graph=step >= 2)
        lg = s.logits().reshape(Bc, W, V)
        # the attention must have followed the cache indirection: logits of hypothesis (b, j) == the model oracle's on
        # that hypothesis' own back-tracked sequence
        n_gen = step + 1
        for b in range(Bc):
            for j in range(W):
                if tr.cum[b, j] < -1e19:  # filler hypotheses (more beams than live candidates)
                    continue
                ref, _ = QO.run_model(qmodel, ids[b:b + 1], lens[b:b + 1], n_gen + 1, feed_ids=seqs[b, j, S:S + n_gen][None])
                np.testing.assert_allclose(lg[b, j], ref[n_gen][0], atol=(8e-2 if sq else 3e-2) * scale,
                                           err_msg=f'step {step} batch {b} beam {j}')
        tr.check(s, lg, first=False)
    out, cum = s.beam_output()
    np.testing.assert_array_equal(out, tr.sequences())
    np.testing.assert_allclose(cum, tr.cum, atol=2e-3)
    assert (np.diff(cum, axis=1) <= 1e-6).all(), 'hypotheses are returned best first'
    # generate() (first step eager, the rest from the captured graph) reproduces the stepped run
    s.setup(Bc, S, NEW, beam_width=W)
    out2 = s.generate(ids, lens, NEW, end_id=-1)
    np.testing.assert_array_equal(out2, out)
    s.close()


def test_beam_width_1_is_greedy():
    cfg, w = synth_model(72)
    S, NEW, B = 8, 6, 2
    lens = np.array([8, 5], np.int32)
    ids = prompts(cfg, B, S, lens, 4)
    s, _ = make_session(cfg, w, 'fp16', 0, ids, lens)
    s.setup(B, S, NEW)
    greedy = s.generate(ids, lens, NEW)
    out, cum = s.beam_output()
    assert cum is None
    np.testing.assert_array_equal(out[:, 0], greedy)
    # and the best beam of a wide search scores at least as well as the greedy path
    s.setup(B, S, NEW, beam_width=4)
    beams = s.generate(ids, lens, NEW)
    assert beams.shape == (B, 4, S + NEW)
    np.testing.assert_array_equal(beams[:, :, :S], np.repeat(ids[:, None], 4, 1))
    s.close()


def test_beam_search_end_id():
    """A finished hypothesis keeps its score, continues with end_id only, and its row is padded with end_id by gather_tree;
    the selection before the end token first appears is unchanged."""
    cfg, w = synth_model(73)
    S, NEW, Bc, W = 9, 12, 1, 3
    lens = np.array([9], np.int32)
    ids = prompts(cfg, Bc, S, lens, 8)
    s, _ = make_session(cfg, w, 'fp16', 0, ids, lens)
    s.setup(Bc, S, NEW, beam_width=W)
    free = s.generate(ids, lens, NEW, end_id=-1)
    end_id = int(free[0, 0, S + 4])  # a token the best hypothesis emits at step 4
    first_hit = min(int(np.where(free[0, j, S:] == end_id)[0][0]) for j in range(W) if (free[0, j, S:] == end_id).any())
    s.setup(Bc, S, NEW, beam_width=W)
    out = s.generate(ids, lens, NEW, end_id=end_id)
    st = s.beam_state()
    _, cum = s.beam_output()
    assert out.shape == (Bc, W, S + NEW)
    for j in range(W):
        row = out[0, j, S:]
        hit = np.where(row == end_id)[0]
        if len(hit):
            assert (row[hit[0]:] == end_id).all()
            assert st['finished'][j] == 1
    assert st['finished'].any(), 'the chosen end token never ended a hypothesis'
    assert (np.diff(cum, axis=1) <= 1e-6).all()
    # tokens emitted before any hypothesis could finish are those of the free run (same candidates, same scores)
    got_prefixes = {tuple(r[S:S + first_hit]) for r in out[0]}
    free_prefixes = {tuple(r[S:S + first_hit]) for r in free[0]}
    if first_hit > 0:
        assert got_prefixes & free_prefixes
    s.close()


def test_generation_session_num_beams():
    """The reference-facing API: GenerationSession.decode with SamplingConfig.num_beams (generation.py:782-997) returns
    [batch, num_beams, max_seq_len]."""
    from tensorrt_llm.runtime import GenerationSession, SamplingConfig
    cfg, w = synth_model(74)
    S, NEW, Bc, W = 8, 5, 2, 2
    lens = np.array([8, 8], np.int32)
    ids = prompts(cfg, Bc, S, lens, 9)
    s, _ = make_session(cfg, w, 'fp16', 0, ids, lens)
    s.setup(Bc, S, NEW, beam_width=W)
    want = s.generate(ids, lens, NEW, end_id=-1)
    gs = GenerationSession.__new__(GenerationSession)
    gs.runtime = s
    gs.batch_size = gs.max_input_length = gs.max_new_tokens = 0
    gs.setup(Bc, S, NEW)
    got = gs.decode(ids, lens, SamplingConfig(end_id=-1, pad_id=2, num_beams=W))
    np.testing.assert_array_equal(np.asarray(got), want)
    with pytest.raises(ValueError):
        gs.setup(1, S, NEW, beam_width=9)  # the device-side beam step takes at most 8 hypotheses per prompt
    # 4 prompts x 4 hypotheses = 16 sequences: more than the 8 rows a generation GEMV launch takes - runs in slabs of 8 since r03
    gs.setup(4, S, NEW, beam_width=4)
    ids4 = prompts(cfg, 4, S, np.array([8, 8, 8, 8], np.int32), 10)
    out4 = gs.decode(ids4, np.array([8, 8, 8, 8], np.int32), SamplingConfig(end_id=-1, pad_id=2, num_beams=4))
    assert np.asarray(out4).shape == (4, 4, S + NEW)
    s.close()
