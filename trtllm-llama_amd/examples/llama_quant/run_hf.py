#!/usr/bin/env python3
"""run_hf.py (T/examples/llama_quant/run_hf.py:22-104): the HF transformers baseline, same prompt / greedy settings /
55-run timing loop, printing `llama-hf-run (mean latency: X sec)`.  Runs on the GPU through torch-ROCm when one is
visible, else on the host CPU (BASELINE.json configs[0])."""
import argparse
import time

import numpy as np
import torch


def parse_arguments():
    p = argparse.ArgumentParser()
    p.add_argument('--max_output_len', type=int, required=True)
    p.add_argument('--log_level', type=str, default='error')
    p.add_argument('--hf_model_location', type=str, default='./tmp/llama/7B/')
    p.add_argument('--tokenizer_dir', type=str, default=None)
    p.add_argument('--input_text', type=str, default='Born in north-east France, Soyer trained as a')
    p.add_argument('--num_beams', type=int, default=1)
    p.add_argument('--num_runs', type=int, default=55)
    p.add_argument('--device', type=str, default=None, choices=[None, 'cpu', 'cuda'])
    return p.parse_args()


def hf_generate(model, ids, max_output_len, num_beams=1, eos_token_id=2, pad_token_id=2, return_logits=False):
    """The reference's generate call (run_hf.py:76-84): greedy (top_k = 1, no sampling) or beam search, EOS = PAD = 2.
    ids: int64 [batch, prompt_len] on the model's device.  Returns the full sequences [batch, prompt_len + new]; with
    return_logits also the raw next-token logits of every step, float32 [new, batch, vocab].  eos_token_id=None keeps
    whatever the model's generation config says (a config without an EOS token never stops early: benchmarks)."""
    kw = dict(max_new_tokens=max_output_len, top_k=1, num_beams=num_beams, do_sample=False, pad_token_id=pad_token_id)
    if eos_token_id is not None:
        kw['eos_token_id'] = eos_token_id
    with torch.no_grad():
        if not return_logits:
            return model.generate(ids, **kw)
        out = model.generate(ids, output_logits=True, return_dict_in_generate=True, **kw)
    return out.sequences, torch.stack([l.float() for l in out.logits])


def main():
    from transformers import LlamaForCausalLM, LlamaTokenizer
    args = parse_arguments()
    device = args.device or ('cuda' if torch.cuda.is_available() else 'cpu')
    tok = LlamaTokenizer.from_pretrained(args.tokenizer_dir or args.hf_model_location, legacy=False)
    model = LlamaForCausalLM.from_pretrained(args.hf_model_location)
    model = (model.half().cuda() if device == 'cuda' else model.float()).eval()
    total = []
    text = ''
    for _ in range(args.num_runs):
        t0 = time.time()
        ids = tok.encode(args.input_text, return_tensors='pt', add_special_tokens=False).to(device)
        out = hf_generate(model, ids, args.max_output_len, args.num_beams)
        text = tok.decode(out[0, ids.shape[1]:].tolist())
        if device == 'cuda':
            torch.cuda.synchronize()
        total.append(time.time() - t0)
    print(f'Input: "{args.input_text}"')
    print(f'Output: "{text}"')
    print(total)
    warm = total[5:] if len(total) > 5 else total
    print(f'llama-hf-run (mean latency: {np.mean(warm)} sec) on {device}')


if __name__ == '__main__':
    main()
