"""The beam-search oracle (oracle/beam_oracle.py) pinned on the CPU: hand-worked vectors, and whole searches against
Hugging Face's independent beam search on a seeded random-init LLaMA (fp32, CPU)."""
import numpy as np
import pytest

from oracle import beam_oracle as BO


def test_step_hand_worked():
    # W = 2, V = 3.  log-probs chosen as exact logs so the arithmetic is checkable by hand.
    lp = np.log(np.array([[0.5, 0.3, 0.2], [0.1, 0.1, 0.8]]))
    cum = np.array([np.log(0.6), np.log(0.4)])
    tok, par, new_cum, fin = BO.beam_step(cum, lp, np.array([False, False]), end_id=2)
    # candidates: 0.6*{.5,.3,.2} = .30 .18 .12 ; 0.4*{.1,.1,.8} = .04 .04 .32  -> best (1,2)=.32 then (0,0)=.30
    np.testing.assert_array_equal(par, [1, 0])
    np.testing.assert_array_equal(tok, [2, 0])
    np.testing.assert_allclose(np.exp(new_cum), [0.32, 0.30])
    np.testing.assert_array_equal(fin, [True, False])
    # next step: hypothesis 0 is finished -> only (0, end_id) at its own score; hypothesis 1 expands
    lp2 = np.log(np.array([[0.9, 0.05, 0.05], [0.6, 0.3, 0.1]]))
    tok, par, new_cum, fin = BO.beam_step(new_cum, lp2, fin, end_id=2)
    np.testing.assert_array_equal(par, [0, 1])
    np.testing.assert_array_equal(tok, [2, 0])
    np.testing.assert_allclose(np.exp(new_cum), [0.32, 0.18])
    np.testing.assert_array_equal(fin, [True, False])


def test_step_ties_go_to_the_lowest_flat_index():
    tok, par, _, _ = BO.beam_step(np.zeros(2), np.zeros((2, 4)), np.array([False, False]), end_id=-1)
    np.testing.assert_array_equal(par, [0, 0])
    np.testing.assert_array_equal(tok, [0, 1])


def test_gather_tree_and_cache_indirection_hand_worked():
    # 1 prompt slot (slot 0), generated slots 1..3, W = 2
    step_ids = np.array([[7, 10, 20, 30], [7, 11, 21, 31]])
    parents = np.array([[0, 0, 1, 0], [0, 0, 0, 0]])
    out = BO.gather_tree(step_ids, parents, last=3, first=1, end_id=-1)
    # hypothesis 0: slot3 tok 30 parent 0 -> slot2 tok 20 parent 1 -> slot1 tok 11
    np.testing.assert_array_equal(out[0], [7, 11, 20, 30])
    # hypothesis 1: slot3 tok 31 parent 0 -> slot2 tok 20 parent 1 -> slot1 tok 11
    np.testing.assert_array_equal(out[1], [7, 11, 20, 31])
    out = BO.gather_tree(step_ids, parents, last=3, first=1, end_id=20)
    np.testing.assert_array_equal(out[0], [7, 11, 20, 20])
    out = BO.gather_tree(step_ids, parents, last=2, first=1, end_id=5)
    np.testing.assert_array_equal(out[:, 3], [5, 5])
    ci = np.zeros((2, 4), np.int32)
    ci = BO.update_cache_indirection(ci, [0, 0], None, 1)       # step after the prompt
    ci = BO.update_cache_indirection(ci, [1, 0], 1, 1)          # tokens 10/11 consumed, K/V at slot 1 of rows 0/1
    np.testing.assert_array_equal(ci[:, :2], [[0, 1], [0, 0]])
    ci = BO.update_cache_indirection(ci, [0, 0], 2, 2)
    np.testing.assert_array_equal(ci[:, :3], [[0, 1, 0], [0, 1, 0]])


@pytest.mark.parametrize('W,seed', [(2, 0), (4, 1), (3, 2)])
def test_whole_search_matches_hf_beam_search(W, seed):
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    V, NEW, S = 96, 6, 5
    m = LlamaForCausalLM(LlamaConfig(hidden_size=64, num_attention_heads=4, num_key_value_heads=4, intermediate_size=128,
                                     vocab_size=V, num_hidden_layers=2, max_position_embeddings=64)).float().eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.ndim == 2:
                p.mul_(4.0)  # wide logits: clear margins between candidates
    prompt = torch.randint(3, V, (1, S), generator=torch.Generator().manual_seed(seed + 10))
    with torch.no_grad():
        hf = m.generate(prompt, num_beams=W, num_return_sequences=W, do_sample=False, max_new_tokens=NEW, min_new_tokens=NEW,
                        length_penalty=0.0, early_stopping=False, eos_token_id=None, pad_token_id=0, output_scores=True,
                        return_dict_in_generate=True)

    def logits_of(seq):
        with torch.no_grad():
            return m(torch.tensor(seq)[None]).logits[0, -1].numpy()

    smax = S + NEW
    step_ids = np.zeros((W, smax), np.int32)
    step_ids[:, :S] = prompt.numpy()
    parents = np.zeros((W, smax), np.int32)
    cum = np.full(W, -1e20)
    cum[0] = 0.0
    fin = np.zeros(W, bool)
    for t in range(NEW):
        seqs = BO.gather_tree(step_ids, parents, last=S + t - 1, first=S, end_id=-1)[:, :S + t]
        lg = np.stack([logits_of(s.tolist()) for s in seqs])
        tok, par, cum, fin = BO.beam_step(cum, lg, fin, end_id=-1)
        step_ids[:, S + t], parents[:, S + t] = tok, par
    mine = BO.gather_tree(step_ids, parents, last=smax - 1, first=S, end_id=-1)
    np.testing.assert_array_equal(mine, hf.sequences.numpy())
    np.testing.assert_allclose(cum, hf.sequences_scores.numpy(), rtol=1e-4, atol=1e-4)
